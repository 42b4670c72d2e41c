/*
 * fil.h -- C ABI of libfil_hip.so: the MI355X (gfx950) feature-interaction hot path.
 *
 * The reference (TIXhjq/ML_Function) is 100% Python/TF2 and has NO native code, so there is no
 * existing FFI to mirror; each entry point below replaces the TF op group that the cited reference
 * lines execute per batch (paths relative to /root/reference/kon/model/ctr_model/layer/).  The
 * Python layer classes in ml_function_amd/layers bind these through ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.cuda storage), row-major contiguous;
 *   - the caller allocates and owns every buffer, including workspaces (sizes from *_workspace_bytes);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no entry point synchronises, allocates or frees (graph-capture safe);
 *   - reductions use a fixed two-stage order: repeated calls on the same inputs are bit-identical;
 *   - return value: 0 on success, negative fil_status on error; fil_last_error() gives the message
 *     of the last failing call on the calling thread.
 */
#ifndef FIL_H_
#define FIL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  FIL_OK = 0,
  FIL_ERR_ARG = -1,         /* NULL pointer / non-positive dim / inconsistent arguments */
  FIL_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed */
  FIL_ERR_WORKSPACE = -3,   /* workspace smaller than *_workspace_bytes() */
  FIL_ERR_UNSUPPORTED = -4  /* shape outside the compiled kernel menu (message says which limit) */
} fil_status;

typedef enum { FIL_F32 = 0, FIL_BF16 = 1 } fil_dtype;

/* ABI version: bumped on EVERY change of an entry point's argument list or semantics.  fil_version() returns the value the
 * library was compiled with; the ctypes binding (ml_function_amd/_lib.py) refuses a library whose value differs from this
 * header's, so a stale prebuilt .so can never be called with shifted arguments. */
#define FIL_ABI_VERSION 215
int fil_version(void);                 /* == FIL_ABI_VERSION of the header the library was built from */
const char* fil_last_error(void);      /* thread-local, never NULL */

/* Opt-in per-kernel timing for bench.py's roofline line (off by default).  Between begin and end every major
 * kernel launch made through this library is bracketed by HIP events recorded ON THE LAUNCH STREAM.
 * fil_profile_end synchronises those events and writes one text line per kernel name into buf:
 *   "<name> <launches> <total_ms> <algorithmic work per launch: flops for MFMA kernels, bytes for streaming> <executed work>\n"
 * (algorithmic = what the reference graph spends on that step; executed = what the kernels really compute -- smaller where an
 * exact algebraic restructuring removes products, e.g. the pair-symmetric first CIN layer; equal otherwise)
 * and returns the number of bytes needed (including the NUL).  Not for use under graph capture. */
int fil_profile_begin(const char* filter);   /* filter: "substr[,substr...]" of kernel names to time, NULL = all */
size_t fil_profile_end(char* buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * A1  FM second order -- replaces InnerLayer.call + FmLayer.call
 *     interactive_layer/interactive_layer.py:59-66,161-170 (C(F,2) tf.multiply + Add + Add).
 *   emb [B,F,K], lin [B,F] or NULL, out [B,K]:  out[b,k] = sum_{i<j} e_i e_j + sum_f lin[b,f].
 *   dtype selects the storage type of emb/out/g/demb (accumulation is always fp32); lin/dlin are fp32.
 *   bwd: demb[b,f,k] = g[b,k] (S[b,k] - e[b,f,k]);  dlin[b,f] = sum_k g[b,k]  (dlin may be NULL).
 */
int fil_fm_fwd(const void* emb, const float* lin, void* out, int B, int F, int K, int dtype, void* stream);
int fil_fm_bwd(const void* emb, const void* g, void* demb, float* dlin, int B, int F, int K, int dtype, void* stream);

/* N3  InnerLayer(use_add=False) pair list (PNN/AFM input), interactive_layer.py:61:
 *   pairs [B, F(F-1)/2, K] in itertools.combinations order; bwd: demb from gpairs. fp32 only. */
int fil_fm_pairs_fwd(const float* emb, float* pairs, int B, int F, int K, void* stream);
int fil_fm_pairs_bwd(const float* emb, const float* gpairs, float* demb, int B, int F, int K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A2  DCN cross network -- replaces CrossLayer.call, interactive_layer.py:275-282, all L layers fused.
 *   x [B,D], w [L,D], b [L,D] (the reference's L tensors [D,1] stacked), y [B,D], s [B,L] (saved dots).
 *   bwd: g [B,D] -> dx [B,D], dw [L,D], db [L,D].
 *   Limit: L <= 16.  D <= 4096 with L <= 6 and 2 L D 4 bytes <= 160 KiB run the register-resident kernels (a sample's row in one
 *   wave's registers, every parameter in LDS, all layers fused); anything else takes the generic two-pass kernels (any D): the closed
 *   form of the recurrence -- L dot products per sample in one pass over its row, then y = c_L x0 + sum_t b_t column by column.
 */
int fil_dcn_fwd(const float* x, const float* w, const float* b, float* y, float* s, int B, int D, int L, void* stream);
size_t fil_dcn_bwd_workspace_bytes(int B, int D, int L);
int fil_dcn_bwd(const float* x, const float* w, const float* b, const float* s, const float* g, float* dx, float* dw,
                float* db, int B, int D, int L, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A3  xDeepFM CIN (north star) -- replaces CIN.call, interactive_layer.py:310-327
 *     (BatchMatMul outer product, 2 transposes, Conv1D 1x1, reduce_sum pooling, Concat, Dense(1)).
 *   x [B,F,K]; W[l] [H_{l-1}*F, H_l] with channel c = h*F+f (H_0 = F); bias[l] [H_l];
 *   dense_w [L*K], dense_b [1] (ignored when output_dim != 1).
 *   fwd outputs: out [B] (output_dim==1; may be NULL otherwise), pooled [B, L*K] (always written),
 *                saved = opaque buffer of fil_cin_saved_bytes bytes that bwd needs (internally: x transposed to
 *                [B*K][F], the feature maps of the layers BELOW the top two as [B*K][128*ceil(H_l/128)], and what the
 *                mode's form of the top two layers keeps: their maps, or the fused tail's [B*K][F+1 columns] and operand
 *                copies, or the quadratic tail's R [B*K][128], T and packed operands; the last layer's map is never
 *                materialised).  Forward and backward must be called with the same shape and mode bits.
 *   bwd: g = dL/dout [B] (output_dim==1) or dL/dpooled [B,L*K];
 *        writes dx [B,F,K], dW[l], dbias[l], ddense_w [L*K], ddense_b [1] (dense grads only if output_dim==1).
 *   mode: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact fp32 products) with the last layer contracted against
 *             wsum_L[c] = sum_n W_L[c,n] (its feature map is only ever sum-pooled, so this is the same function at 1/H_L of
 *             the flops) and, for L >= 3, the layer below it contracted against [1 | wsum_L] (its map is only ever sum-pooled
 *             or fed to the last layer: F+1 observable columns instead of H_{L-1}; the "fused tail", csrc/cin_tail.h);
 *         1 = fp32 MFMA with every layer through the general GEMM kernels (validation / comparison).
 *         + FIL_CIN_BF16X3 (2; ABI 214): the LABELLED reduced-operand mode ("bf16 operands + fp32 accumulate").  The three
 *           GEMM launches of the merged quadratic tail run on split-bf16 operands (csrc/cin_qsplit.h): every fp32 operand as three bf16
 *           pieces (an exact cut), every product as six v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the fp32-equivalent chain
 *           a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 (dropped terms < 2^-24 relative) on the bf16 matrix pipe.  Holds the parity bars
 *           of mode 0 (tests run both side by side); not bit-identical to it (other summation order), and an infinite input gives
 *           NaN where the exact chain gives inf.  Selects something only where those kernels exist (three layers on the merged
 *           quadratic tail, F in the kernels' menu, H_1 <= 128): elsewhere the call runs the exact kernels.  Never the default;
 *           bench.py reports it beside the exact-fp32 headline, not as it.  (Bit 2 of ABI <= 209 was the same idea on the round-2
 *           layer structure; ABI 210-213 answered it with FIL_ERR_UNSUPPORTED.)
 *         + FIL_CIN_X_TRANSPOSED (16), forward and backward alike: `x` is given transposed, [B*K][F] row-major (x_t[(b*K+k)*F+f]
 *           = x[b,f,k], as written by fil_embed_gather_xt): no input transpose, saved's own copy of it stays unused.  dx is
 *           still returned as [B,F,K].
 *         + FIL_CIN_NOTAIL (32): mode 0 without the fused tail (last layer through wsum_L only: the round-2 path);
 *           FIL_CIN_TAIL_ALWAYS (64): the fused tail whenever it is defined (L >= 3, F <= 62), also where it saves nothing
 *           (by default it is used when F+2 columns padded to 16 are at most 3/4 of H_{L-1}); both exist for tests / comparison.
 *         + FIL_CIN_NOQTAIL (256): for THREE layers (and F + 2 <= 64, H_1 <= 128, 3F + 1 <= the widest feature-map row) mode 0 goes one
 *           step further than the fused tail: pool_L = sum_h x^1[h] (x^T T_h x) + <x, c> + const is a quadratic form in x for every
 *           feature map h of the FIRST layer, so the top two layers run as a product over UNORDERED field pairs with the first layer's
 *           pair-symmetric GEMM kernels -- F(F+1)/2 x H_1 products per row, half of the fused tail's, no column padding
 *           (csrc/cin_qtail.h).  Used above 16,384 rows (B*K; smaller batches are launch-latency bound and keep the fused tail,
 *           FIL_CIN_TAIL_ALWAYS lifts that rule).  This bit keeps three-layer nets on the F+1-column fused tail (tests / comparison).
 *         + FIL_CIN_NOQMERGE (512): the quadratic tail normally runs THREE GEMM launches (csrc/cin_qmerge.h): the forward
 *           [x^1 | R] = pairs(x) [W_1 | T] with 256 output columns and all three sum-pools in its epilogue, the weight gradients
 *           pairs(x)^T [G^1 | dP_L x^1] with 256 output columns, and the data gradients as two passes of one launch over one dX image.
 *           This bit keeps round 3's two launches per direction (tests / comparison).  Same function up to summation order.
 *         + FIL_CIN_NOKSPLIT (128): small batches (B*K <= 16,384 rows) give each block of 32 rows to the FOUR waves of a workgroup,
 *           which split the reduction between them (strong-scaling shards: without it the row-parallel kernels stop getting faster
 *           below one row block per SIMD); this bit keeps one wave per row block.  Same function up to summation order.
 *         + FIL_CIN_MB2 (4) / FIL_CIN_NOSYM (8): per-call launch-shape overrides (64-row waves in the row-parallel kernels,
 *             i.e. the launch configuration large batches get by themselves; symmetric first-layer kernels off).  Same
 *             function up to summation order; they exist so that tests can reach every instantiation at small sizes.
 *   grad_ready_events (bwd; may be NULL): L+1 hipEvent_t handles (entries may be NULL).  [l] is recorded on `stream` once
 *       dW[l] and dbias[l] are final, [L] once the dense head's gradients are -- from the top layer down, each before the
 *       data-gradient kernel of its layer starts -- so a data-parallel caller can start the all-reduce of a layer's
 *       gradients on another stream while the rest of the backward is still running.
 *       fil_cin_grad_ready_points(.., mode, point) tells WHERE in the backward each slot is recorded: point[l] (l = 0..L) = ordinal of
 *       the recording point, 0 first; slots with the same ordinal are recorded together (fused tail: the two top layers), so a
 *       caller reduces them with ONE collective.  Returns the number of distinct points, or a (negative) error code.
 *   Limits: F <= 64, H_l <= 256, L <= 8, B*K <= 2^28.
 */
size_t fil_cin_saved_bytes(int B, int F, int K, int L, const int* H);
size_t fil_cin_fwd_workspace_bytes(int B, int F, int K, int L, const int* H);
size_t fil_cin_bwd_workspace_bytes(int B, int F, int K, int L, const int* H);
int fil_cin_grad_ready_points(int B, int F, int K, int L, const int* H, int mode, int* point);
int fil_cin_fwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                const float* dense_b, float* out, float* pooled, float* saved, int B, int F, int K, int L,
                const int* H, int output_dim, int mode, void* workspace, size_t workspace_bytes, void* stream);
int fil_cin_bwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                const float* pooled, const float* saved, const float* g, float* dx, float* const* dW,
                float* const* dbias, float* ddense_w, float* ddense_b, int B, int F, int K, int L, const int* H,
                int output_dim, int mode, void* const* grad_ready_events, void* workspace, size_t workspace_bytes,
                void* stream);

/* ---------------------------------------------------------------------------------------------
 * A4  AutoInt interacting layer -- replaces MultHeadAttentionLayer.call + ProductAttentionLayer.call
 *     (behavior_layer/behavior_layer.py:292-311,356-377) and DnnLayer's Add + ReLU
 *     (core_layer/core_layer.py:204-216), fused; the [H,B,F,F] score tensor is never materialised.
 *   x [B,F,K]; Wq, Wk, Wr [K,H,A] (V is projected with Wk, as the reference does; Wr may be NULL = use_res off);
 *   gamma, beta [A] (NULL = use_ln off); eps = 1e-3 for Keras parity; scale = 1/sqrt(A) (use_scale) or 1.
 *   fuse_relu != 0 (the DnnLayer(res_unit=1, other_dense=[layer]) wrapper AutoInt uses):
 *       y [H,B,F,A] = relu(x Wr + LN(sigmoid(scale * q k^T) k)),  res_out ignored.
 *   fuse_relu == 0 (MultHeadAttentionLayer.call as a stand-alone layer, which returns [atten_v, res]):
 *       y = LN(sigmoid(scale * q k^T) k),  res_out [H,B,F,A] = x Wr (may be NULL).
 *   bwd: dy [H,B,F,A] (and, when fuse_relu == 0 and Wr != NULL, dres_in [H,B,F,A] = gradient of res_out)
 *        -> dx [B,F,K], dWq, dWk, dWr [K,H,A], dgamma, dbeta [A].
 *   av_out [H,B,F,A] + rstd_out [H,B,F] (fwd, optional, both or neither; used with gamma): the LayerNorm input saved for the
 *       backward, NORMALISED -- (av - mean) / sqrt(var + eps) -- with the rows' 1 / sqrt(var + eps) beside it, so that the
 *       backward's LayerNorm gradient does not derive the statistics of every row again (av_saved + rstd_saved).  The fused
 *       ReLU mask is y > 0: pass the forward's y back as y_saved.  If any of them is NULL the backward first re-runs the
 *       forward into its workspace (fil_attn_bwd_workspace_bytes(.., have_saved = 0)); same gradients bit for bit.
 *   x_chunk: 0 = x (and dx) are [B,F,K].  c > 0 = head-major [K/c][B][F][c], i.e. the [H',B,F,A'] output of a previous
 *       interacting layer read in place as its head-concat [B,F,H'*A'] (ESULayer's convention, behavior_layer.py:973):
 *       a stack of interacting layers (BASELINE config 5: 3 layers) needs no transposes, and dx IS the dy of the layer below.
 *   The backward is one kernel (one pass over the F x F scores): per-workgroup partial sums of dW* / dgamma / dbeta are
 *   reduced afterwards in a fixed order (bit-identical repeats).
 *   precision: FIL_PREC_F32 = every matrix product on the exact fp32 MFMA (1e-5 parity with the reference);
 *       FIL_PREC_F16_MFMA = BASELINE config 5 ("fp16 MFMA QK^T V"): the operands of every matrix product (projections,
 *       scores, weighted sums and their gradients) are rounded to fp16 and multiplied on v_mfma_f32_16x16x16_f16 with
 *       fp32 accumulation; sigmoid, LayerNorm, residual, ReLU, all tensors in memory and all reductions stay fp32.
 *       Parity of that mode is ~1e-3 (tests state 5e-3 / 2e-2); values beyond the fp16 range (65504) overflow.
 *   Limits: K <= 64, A <= 16, H <= 8, F <= 512 (and the LDS footprint <= 160 KiB).
 */
enum fil_cin_mode_bits { FIL_CIN_GENERAL = 1, FIL_CIN_BF16X3 = 2, FIL_CIN_MB2 = 4, FIL_CIN_NOSYM = 8, FIL_CIN_X_TRANSPOSED = 16,
                         FIL_CIN_NOTAIL = 32, FIL_CIN_TAIL_ALWAYS = 64, FIL_CIN_NOKSPLIT = 128, FIL_CIN_NOQTAIL = 256, FIL_CIN_NOQMERGE = 512 };
enum fil_precision { FIL_PREC_F32 = 0, FIL_PREC_F16_MFMA = 1 };
size_t fil_attn_fwd_workspace_bytes(int B, int F, int K, int H, int A);
size_t fil_attn_bwd_workspace_bytes(int B, int F, int K, int H, int A, int have_saved);
int fil_attn_fwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                 const float* beta, float* y, float* res_out, float* av_out, float* rstd_out, int B, int F, int K, int H,
                 int A, float scale, float eps, int fuse_relu, int precision, int x_chunk, void* workspace,
                 size_t workspace_bytes, void* stream);
int fil_attn_bwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                 const float* beta, const float* dy, const float* dres_in, const float* y_saved, const float* av_saved,
                 const float* rstd_saved, float* dx, float* dWq, float* dWk, float* dWr, float* dgamma, float* dbeta, int B, int F, int K, int H,
                 int A, float scale, float eps, int fuse_relu, int precision, int x_chunk, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ProductAttentionLayer.call([q, k, v], mask) as a stand-alone layer (behavior_layer.py:292-311): separate q / k / v, any
 * mask.  (AutoInt's own path is fil_attn_*: projections, attention, LayerNorm and residual fused.)
 *   q [N,Fq,A], k [N,Fk,A], v [N,Fk,Av] -> out [N,Fq,Av] = sigmoid(scale * q k^T + mask * (-1e5)) v     (fp32 MFMA)
 *   N = the flattened leading axes (e.g. heads x batch); scale = 1/sqrt(A) for use_scale, else 1.
 *   mask (may be NULL) [mask_period, Fq, Fk], item n uses mask[n % mask_period]: the additive mask of mask_mod == 2
 *   (:303-306).  mask_mod == 1 (:300-302, the scores right-multiplied by a mask matrix M) is (q k^T) M = q (M^T k)^T:
 *   the caller passes k' = M^T k and no mask.
 *   bwd: dout [N,Fq,Av] -> dq, dk, dv.  Limits: A, Av <= 64; LDS footprint 80*(ceil(A/16)+ceil(Av/16))*max(Fq,Fk) bytes <= 160 KiB.
 */
int fil_pattn_fwd(const float* q, const float* k, const float* v, const float* mask, float* out, int N, int Fq, int Fk, int A,
                  int Av, float scale, int mask_period, void* stream);
int fil_pattn_bwd(const float* q, const float* k, const float* v, const float* mask, const float* dout, float* dq, float* dk,
                  float* dv, int N, int Fq, int Fk, int A, int Av, float scale, int mask_period, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N1  SparseEmbed field-index work -- replaces the F Embedding lookups of SparseEmbed.call,
 *     interactive_layer.py:225-242, emitting the packed [B,F,K] layout directly (bit-exact row copies).
 *   table: all F tables concatenated [sum_f V_f, K]; offsets [F] = first row of field f (int64);
 *   sizes [F] = V_f (int64; NULL = ids are trusted); idx [B,F] (int64, per-field local ids).  out [B,F,K].
 *   An id outside [0, V_f) gives a ZERO output row (Keras' Embedding on a GPU) and is counted in *oob_count (device int,
 *   may be NULL; the caller zeroes it) -- never a read of the neighbouring field's table; its gradient is dropped.
 *   Gradient, deterministic (default in the Python layer):
 *     fil_embed_row_ids      row_ids[b*F+f] = offsets[f] + idx[b,f], or -1 for out-of-range ids and frozen fields
 *                            (frozen [F] bytes, may be NULL: sparseFea.is_trainable = False);
 *     (the caller sorts row_ids stably -> perm, and takes the run starts of the sorted ids -> rows_out [U], starts [U+1])
 *     fil_embed_segment_sum  values[u,:] = sum_{j in [starts[u], starts[u+1])} g[perm[j], :] in sorted order with a fixed
 *                            lane tree; rows_out[u] < 0 is skipped; dtable != NULL additionally stores the row sums into
 *                            dtable[rows_out[u], :] (unique rows: plain stores).  values / dtable: either may be NULL.
 *   fil_embed_scatter_add: the fp32-atomic alternative, dtable[offsets[f]+idx[b,f], :] += g[b,f,:] (dtable pre-zeroed;
 *                            the order of additions into a row hit several times is NOT fixed).
 */
int fil_embed_gather(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, float* out,
                     int* oob_count, int B, int F, int K, void* stream);
/* the same with the block's storage type chosen (out_dtype FIL_F32 / FIL_BF16: the rows leave rounded to bf16 -- what a bf16 model
 * would otherwise do to the block in a cast launch of its own; the table stays fp32) */
int fil_embed_gather_dt(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, void* out,
                        int* oob_count, int B, int F, int K, int out_dtype, void* stream);
/* the same gather emitting BOTH the packed block out [B,F,K] and its transposed rows out_t [B*K][F] (out_t[(b*K+k)*F+f] =
 * out[b,f,k]) -- the layout the CIN kernels read: hand out_t to fil_cin_fwd / fil_cin_bwd as `x` with FIL_CIN_X_TRANSPOSED and
 * the consumer's input transpose is gone (gather fused into the consumer). */
int fil_embed_gather_xt(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, float* out,
                        float* out_t, int* oob_count, int B, int F, int K, void* stream);
int fil_embed_scatter_add(const int64_t* offsets, const int64_t* sizes, const int64_t* idx, const float* g, float* dtable,
                          int B, int F, int K, void* stream);
int fil_embed_row_ids(const int64_t* offsets, const int64_t* sizes, const unsigned char* frozen, const int64_t* idx,
                      int64_t* row_ids, int B, int F, void* stream);
int fil_embed_segment_sum(const float* g, const int64_t* perm, const int64_t* starts, const int64_t* rows_out, float* values,
                          float* dtable, long U, int K, void* stream);
/* row ids AND their sort in one launch, for fil_embed_run_sum: field f's B entries sorted by (row id, position) -- a stable sort
 * within the field -- land in sorted_ids / perm [f*B, (f+1)*B) (perm[j] = b*F + f); skipped entries (-1) lead each field's segment.
 * Equal row ids are adjacent and in position order, which is all the run sum needs (rows of different fields cannot be equal), but
 * the list as a whole is sorted only if offsets[] ascends.  B <= 8192 (one workgroup sorts a field in LDS), ids < 2^32 - 1.
 * max_vocab: an upper bound of every field's vocabulary (the concatenated table's row count will do), 0 = unknown: when
 * (max_vocab + 1) * 2^ceil(log2 B) fits 32 bits the sort runs on 32-bit composites (half the LDS traffic). */
int fil_embed_sort_fields(const int64_t* offsets, const int64_t* sizes, const unsigned char* frozen, const int64_t* idx,
                          int64_t* sorted_ids, int64_t* perm, int B, int F, int64_t max_vocab, void* stream);
/* the same sums without any data-dependent size (HIP-graph capturable): sorted_ids [R] = the stably sorted row ids, perm [R]
 * the sorting permutation; the run of every distinct id >= 0 is summed in sorted order into the (pre-zeroed) dense dtable. */
int fil_embed_run_sum(const float* g, const int64_t* perm, const int64_t* sorted_ids, float* dtable, long R, int K, void* stream);
/* the same for a gradient block stored as g_dtype (FIL_F32 / FIL_BF16: converted on load, summed in fp32 into the fp32 dtable) */
int fil_embed_run_sum_dt(const void* g, const int64_t* perm, const int64_t* sorted_ids, float* dtable, long R, int K, int g_dtype,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * N2  The score head and the loss behind the interaction layers (all tensors fp32, n = batch rows).
 *   ScoreLayer(use_add=True).call (kon/model/ctr_model/layer/core_layer/core_layer.py:58-84): keras Add over the [B,1] parts,
 *   then sigmoid.  fil_score_add_sigmoid_fwd: out[i] = sigmoid(((a[i] + b[i]) + c[i]) + d[i]) -- left to right like the
 *   reference's Add; b, c, d may be NULL.  _bwd: dsum[i] = dout[i] * out[i] * (1 - out[i]), the gradient of EVERY part.
 *   fil_bce_mean_fwd: binary cross-entropy on probabilities as the reference compiles it (example/ctr_example/un_seq.py:61,
 *   model.compile(loss=tf.losses.binary_crossentropy); TensorFlow 2.1 keras/backend.py): pc = clip(p, eps, 1 - eps),
 *   loss[0] = mean(-(y log(pc + eps) + (1 - y) log(1 - pc + eps))); dp (may be NULL) [n] = d loss / d p (0 where the clip is
 *   active).  Keras' eps is 1e-7.  One workgroup, fixed summation order: repeats are
 *   bit-identical.  Meant for training batches (n up to ~1e5); n >= 1.
 */
int fil_score_add_sigmoid_fwd(const float* a, const float* b, const float* c, const float* d, float* out, int n, void* stream);
int fil_score_add_sigmoid_bwd(const float* out, const float* dout, float* dsum, int n, void* stream);
int fil_bce_mean_fwd(const float* p, const float* y, float eps, float* loss, float* dp, int n, void* stream);

/* MergeScoreLayer.call (core_layer/core_layer.py:86-100): StackLayer concat of the n_parts (1..4) tensors parts[i] [B, widths[i]] ->
 * Dense(O <= 8 units, softmax): out [B, O] = softmax(concat(parts) W + bias), W [D = sum widths, O] row-major (the Keras Dense kernel),
 * bias [O].  DeepFM / DCN / Wide&Deep end in it (model/models.py:87,104).  One launch forward, two backward, the parts read where they lie (no
 * concatenated copy).  dtypes[i] = storage type of parts[i] and of dparts[i] (FIL_F32 or FIL_BF16: a model under bf16 autocast hands
 * a bf16 FM output next to an fp32 MLP output); W, bias, out, dout, dW, db are fp32 and so is the arithmetic.
 *   bwd: dz = out (dout - <dout, out>); dparts[i] [B, widths[i]] = dz W_i^T (entries / the array may be NULL: not wanted);
 *        dW [D, O] = concat(parts)^T dz, db [O] = column sums of dz -- block partials summed in block order by a second, tiny launch:
 *        deterministic.  workspace: fil_merge_softmax_bwd_workspace_bytes(B, D, O) bytes. */
/* The dense layers of the zoo's MLPs (DnnLayer / HiddenLayer / Dense, core_layer/core_layer.py:102-118,201-226) at batch-sized M and a
 * few hundred columns: C [M][N] = op(A) [M][K] . op(B) [K][N] (+ bias[n] (epilogue 1), then ReLU (2)), fp32, exact
 * v_mfma_f32_32x32x2_f32 chains, deterministic (a small output with a long reduction -- dW = x^T dz -- is split over k and the slices
 * are summed in order by a second launch: that is what the workspace is for; a call with an epilogue is never split).
 *   trans_a = 0: A is [M][K] row-major, leading dimension lda >= K;  1: A is [K][M], lda >= M   (dW = x^T dz: A = x, trans_a = 1)
 *   trans_b = 0: B is [K][N] row-major, ldb >= N;                    1: B is [N][K], ldb >= K   (dx = dz W^T: B = W [in][out], trans_b = 1)
 * Any M, N, K; 16-byte operand loads where base and leading dimension allow, element-wise otherwise. */
size_t fil_gemm_f32_workspace_bytes(int M, int N, int K);
int fil_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                 int trans_b, int epilogue, void* workspace, size_t workspace_bytes, void* stream);

/* Backward half of Dense + bias + ReLU (the zoo's MLP layers, DnnLayer core_layer/core_layer.py:102-118,201-226; the layer's GEMMs stay
 * library GEMMs): dz[b,n] = y[b,n] > 0 ? dy[b,n] : 0 (y = the layer's OUTPUT) and dbias[n] = sum_b dz[b,n] in ONE pass over [B, N]
 * (torch: threshold_backward, then a column reduce).  y, dy, dz in `dtype` storage (FIL_F32 / FIL_BF16), dbias fp32; deterministic (block
 * partials summed in block order by a second, tiny launch). */
size_t fil_relu_bias_bwd_workspace_bytes(int B, int N);
int fil_relu_bias_bwd(const void* y, const void* dy, void* dz, float* dbias, int B, int N, int dtype, void* workspace, size_t workspace_bytes,
                      void* stream);
size_t fil_merge_softmax_bwd_workspace_bytes(int B, int D, int O);
int fil_merge_softmax_fwd(const void* const* parts, const int* widths, const int* dtypes, int n_parts, const float* W, const float* bias,
                          float* out, int B, int O, void* stream);
int fil_merge_softmax_bwd(const void* const* parts, const int* widths, const int* dtypes, int n_parts, const float* W, const float* out,
                          const float* dout, void* const* dparts, float* dW, float* db, int B, int O, void* workspace, size_t workspace_bytes,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FIL_H_ */
