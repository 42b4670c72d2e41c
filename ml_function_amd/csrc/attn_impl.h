// The AutoInt interacting layer's kernels, launchers and entry points, compiled as three translation units (FIL_ATTN_PART: 0 = attn.hip: the
// forward, the reduce kernel and the C entry points; 1 = attn_bwd_f16.hip, 2 = attn_bwd_f32.hip: the backward's instantiations per precision).
#pragma once
// A4  AutoInt interacting layer (multi-head field attention) for gfx950.
//
// Replaces MultHeadAttentionLayer.call + ProductAttentionLayer.call (behavior_layer.py:292-311,356-377) and the
// Add + ReLU of the DnnLayer wrapper (core_layer.py:204-216).  Reference quirks kept: the "softmax" is a sigmoid,
// V is projected with key_w (so K == V), LayerNorm eps is Keras' 1e-3, output is head-major [H,B,F,A].
//
// What bounds this layer (B=4096, F=200, H=4, A=16): every [H,B,F,A] tensor is 210 MB and a pass over the F x F scores
// is 655 M sigmoids (v_exp_f32 + v_rcp_f32, quarter rate), so the design minimises (a) HBM passes over [H,B,F,A]
// tensors and (b) sigmoid passes; the matrix pipe is nowhere near busy.  Two kernels:
//
//   attn_fwd_kernel   one workgroup = one sample, one wave = one head.  x is staged once for all heads; the wave projects
//                     its k (== v) -- into registers, both orientations, for up to 13 key tiles in the f16 mode, else into a
//                     private LDS image -- and walks the 16x16 score tiles:
//                         S'[key][query] = sigmoid(scale k_t q_i^T)      accumulator: key on (lane>>4, reg), query on lane&15
//                         av^T[a][query] += k_t^T S'                     the S' accumulator IS the B operand, no LDS trip
//                     av^T leaves the MFMA as [query on the lane][4 consecutive a] = one 16-byte store per lane, so y
//                     and the saved rows are written as whole 1-KB rows; LayerNorm + residual + ReLU in the epilogue (the
//                     rows kept for the backward are the NORMALISED ones, with their 1/sigma beside them).
//   attn_bwd_kernel   ONE pass over the scores for the whole backward (the round-1 code made three).  Same
//                     decomposition; the wave keeps dk of ALL its key tiles in accumulators (13 x 4 registers at F=200)
//                     while it loops over the query blocks:
//                         S [query][key] , dS = dav_i k_t^T ; dP = dS S (1-S)
//                         dk_t^T[a][key] += q_i^T dP + dav_i^T S          S and dP accumulators are the B operands
//                         dq_i^T[a][query] += k_t^T dP^T                  dP crosses LDS once (8-byte write + transposing
//                                                                         ds_read_b64_tr_b16 in the f16 mode)
//                     The LayerNorm/ReLU backward (from the saved normalised rows, 1/sigma and y) is the prologue of each query block, and
//                     the projection gradients are folded in: dW* accumulate in registers across the samples of a
//                     persistent workgroup, dx of the heads is summed in an LDS tile (fixed order) and written once.
//                     HBM traffic: x, av, y, dy in; dx out -- no dav/dq/dk/dres round trips.  Where the LDS footprint
//                     allows one workgroup per CU only (K = 64, the f32 mode at large F) a head gets TWO waves that take
//                     alternate query blocks (WPH = 2; nothing is computed twice; wave w = head w mod H, so the pair shares a SIMD).
//
// precision F32: every product on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains; 1e-5 parity mode), x fragments read from
// global/L2.  F16_MFMA (BASELINE config 5): operands rounded to fp16 once when they enter LDS / registers, products on
// v_mfma_f32_16x16x16_f16 with fp32 accumulation; sigmoid, LayerNorm, residual, ReLU, every reduction and every tensor
// in HBM stay fp32.
//
// MFMA 16x16 maps: lane l supplies A[i=l&15][k=4(l>>4)+s], B[k=4(l>>4)+s][j=l&15], s=0..3; D reg r = D[4(l>>4)+r][l&15].
// "row fragment" of a [16 x 16] tile T: lane (g,c) holds T[c][4g..4g+3]; "column fragment": lane (g,c) holds T[4g..4g+3][c].
#include "common.h"
#include <algorithm>
#include <mutex>
#include <type_traits>

namespace fil {

constexpr int kMaxNC = 4;      // K <= 64 (NC = ceil(K/16) chunks of 16 along the projection's reduction)
constexpr int kMaxHeads = 8;   // one wave per head (two in the backward's WPH = 2 form, H <= 4)

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

template <bool F16>
struct Prec;
template <>
struct Prec<true> {
  typedef _Float16 Elem;
  typedef f16x4 Op;
  static constexpr int RS = 16;   // LDS tile row stride (elements): 32-byte rows, 8-byte chunks XOR-swizzled
};
template <>
struct Prec<false> {
  typedef float Elem;
  typedef f32x4 Op;
  static constexpr int RS = 20;   // 80-byte rows: conflict-free b128 row reads and strided b32 column reads
};

// ---- one MFMA "step": 4 reduction indices per lane group ------------------------------------------------------------
template <bool F16>
__device__ __forceinline__ f32x4 mma(typename Prec<F16>::Op a, typename Prec<F16>::Op b, f32x4 c) {
  if constexpr (F16) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
}

template <bool F16>
__device__ __forceinline__ typename Prec<F16>::Op to_op(f32x4 v) {
  if constexpr (F16) {
    return __builtin_convertvector(v, f16x4);
  } else {
    return v;
  }
}

template <bool F16>
__device__ __forceinline__ f32x4 from_op(typename Prec<F16>::Op v) {
  if constexpr (F16) {
    return __builtin_convertvector(v, f32x4);
  } else {
    return v;
  }
}

// ---- LDS tile images: [rows][16] elements ------------------------------------------------------------------------------
// f16: 32-byte rows; the four 8-byte chunks of row r are stored at chunk ^ ((r >> 2) & 3).  Conflict-free for all three
// access shapes: 8-byte row reads (a 32-lane half = 16 rows x 2 chunks over 64 banks: rows r and r+8 share their bank
// group and get chunk sets {f, f^1} vs {f^2, f^3}), transposing reads (8 rows x 4 chunks = 64 different dwords), and
// 8-byte row writes (16 lanes = 16 rows x 1 chunk over 32 banks: rows r, r+4, r+8, r+12 collide in 8r mod 32 and take
// four different chunks) -- the plain layout made the writes 4-way (SQ_LDS_BANK_CONFLICT 27 % of the LDS cycles).
__device__ __forceinline__ int swz16(int row) { return (row >> 2) & 3; }

template <bool F16>
__device__ __forceinline__ typename Prec<F16>::Op row_read(const typename Prec<F16>::Elem* img, int row, int g) {
  if constexpr (F16) {
    return *reinterpret_cast<const f16x4*>(img + row * 16 + 4 * (g ^ swz16(row)));
  } else {
    return *reinterpret_cast<const f32x4*>(img + row * 20 + 4 * g);
  }
}

template <bool F16>
__device__ __forceinline__ void row_write(typename Prec<F16>::Elem* img, int row, int g, typename Prec<F16>::Op v) {
  if constexpr (F16) {
    *reinterpret_cast<f16x4*>(img + row * 16 + 4 * (g ^ swz16(row))) = v;
  } else {
    *reinterpret_cast<f32x4*>(img + row * 20 + 4 * g) = v;
  }
}

// column fragment of the 16-row block that starts at row0: element s = img[row0 + 4g + s][lane & 15].
// f16: ds_read_b64_tr_b16 -- lane 4q+p of a 16-lane group supplies the address of (row q, chunk p) of the group's
// 4 x 16 block and receives column (lane & 15) of its 4 rows.  EXEC must be all ones (call from wave-uniform code only).
template <bool F16>
__device__ __forceinline__ typename Prec<F16>::Op tr_read(const typename Prec<F16>::Elem* img, int row0, int lane) {
  if constexpr (F16) {
    const int li = lane & 15;
    const int row = row0 + 4 * (lane >> 4) + (li >> 2);
    const _Float16* p = img + row * 16 + 4 * ((li & 3) ^ swz16(row));
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(f16x4, r);
  } else {
    const float* p = img + (row0 + 4 * (lane >> 4)) * 20 + (lane & 15);
    return f32x4{p[0], p[20], p[40], p[60]};
  }
}

// v[lane] + v[lane ^ 16] in every lane with the gfx950 VALU row swap (v_permlane16_swap exchanges the odd rows of its first
// operand with the even rows of its second), instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ float lane_rows_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// sum over the four lanes {c, c+16, c+32, c+48} (the four 4-element pieces of one fragment row), result in all of them.
// One v_mfma_f32_16x16x4_f32 with A = 1: D[i][j] = sum_k B[k][j], and a lane's B element is B[k = lane >> 4][j = lane & 15] -- an
// exact fp32 chain ((v0 + v1) + v2) + v3 in lane-group order.  The cross-lane form (v_permlane16_swap + v_permlane32_swap, each
// fenced by s_nops) was ten vector-issue slots per sum; the LayerNorm backward makes four per query block, and the attention
// kernels are bound by vector issue while their matrix pipe idles.
__device__ __forceinline__ float groups_sum(float v) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)[0];
}

// sum over the 16 lanes of a row (lanes 16g .. 16g+15), result in all of them: DPP only, no LDS crossbar
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);  // row_mirror
  return v;
}

__device__ __forceinline__ float sigmoid_from_neg_log2(float t) {   // t = -log2(e) * score  ->  1 / (1 + exp(-score))
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

// makes the compiler wait HERE for a pending load of v (an empty asm that "uses" the registers)
__device__ __forceinline__ void settle4(f32x4& v) { asm volatile("" : "+v"(v)); }

struct AttnDims {
  int B, F, K, H, A;
  int nblk;   // ceil(F/16)
  int FP;     // 16*nblk
  int NC;     // ceil(K/16)
  int xcw;    // x chunk width: K for the plain [B,F,K] layout, c for the head-major layout [K/c][B][F][c]
  int xcs;    // x chunk stride in elements (B*F*c; 0 for the plain layout)
  int xrcp;   // ceil(2^16 / xcw) (0 for the plain layout): kin / xcw == (kin * xrcp) >> 16 for kin < 64
};

// element offset of x[b, f, kin] (the host guarantees B*F*K < 2^29, so offsets and byte offsets fit 32 bits)
__device__ __forceinline__ int x_off(const AttnDims& d, int b, int f, int kin) {
  const int ch = (kin * d.xrcp) >> 16;
  return ch * d.xcs + (b * d.F + f) * d.xcw + (kin - ch * d.xcw);
}

// raw buffer access: a lane whose byte offset is >= the descriptor's size reads 0 / stores nothing, so masked lanes cost
// neither a branch nor a select (kOOB is added to the offset of lanes that must not touch memory)
constexpr int kOOB = 0x40000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(bytes < 0x3fffffffL ? bytes : 0x3fffffffL), 0x00020000);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, byte_off, 0, 0);
}

// ---- x fragments --------------------------------------------------------------------------------------------------------
// XL = true (f16 mode): x[b] converted once into LDS images xs[NC][FP][16] (zero padded), fragments are LDS reads.
// XL = false: fragments straight from global (L1/L2; x[b] is 13-51 KB and shared by the heads), converted on the fly.
template <bool F16, bool XL>
struct XSrc;

template <>
struct XSrc<true, true> {
  const _Float16* xs;
  int FP;
  __device__ __forceinline__ void set_sample(const AttnDims&, int) {}
  // row fragment: element s = x[16 blk + (lane&15)][16 ch + 4g + s]
  __device__ __forceinline__ f16x4 row(int blk, int ch, int lane) const {
    return row_read<true>(xs + ch * FP * 16, 16 * blk + (lane & 15), lane >> 4);
  }
  // column fragment: element s = x[16 blk + 4g + s][16 ch + (lane&15)]
  __device__ __forceinline__ f16x4 col(int blk, int ch, int lane) const { return tr_read<true>(xs + ch * FP * 16, 16 * blk, lane); }
};

template <bool F16>
struct XSrc<F16, false> {
  __amdgpu_buffer_rsrc_t rx;   // the whole x tensor
  AttnDims d;
  int b;
  bool vec;   // xcw % 4 == 0 && K % 4 == 0: four consecutive kin are one aligned 16-byte piece
  __device__ __forceinline__ void init(const float* x, const AttnDims& dd) {
    d = dd;
    rx = make_rsrc(x, (long)dd.B * dd.F * dd.K * 4);
    vec = (dd.xcw & 3) == 0 && (dd.K & 3) == 0;
  }
  __device__ __forceinline__ void set_sample(const AttnDims&, int bb) { b = bb; }
  __device__ __forceinline__ typename Prec<F16>::Op row(int blk, int ch, int lane) const {
    const int f = 16 * blk + (lane & 15), k0 = 16 * ch + 4 * (lane >> 4);
    f32x4 v;
    if (vec) {
      v = buf_load4(rx, (f < d.F && k0 < d.K) ? 4 * x_off(d, b, f, k0) : kOOB);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = buf_load1(rx, (f < d.F && k0 + s < d.K) ? 4 * x_off(d, b, f, k0 + s) : kOOB);
    }
    return to_op<F16>(v);
  }
  __device__ __forceinline__ typename Prec<F16>::Op col(int blk, int ch, int lane) const {
    const int f0 = 16 * blk + 4 * (lane >> 4), kin = 16 * ch + (lane & 15);
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = buf_load1(rx, (kin < d.K && f0 + s < d.F) ? 4 * x_off(d, b, f0 + s, kin) : kOOB);
    return to_op<F16>(v);
  }
};

// idx / per_ch for idx < NC * per_ch without an integer division (a division by a run-time value is ~25 vector instructions)
template <int NC>
__device__ __forceinline__ int chunk_of(int idx, int per_ch) {
  int ch = 0;
#pragma unroll
  for (int k = 1; k < NC; ++k) ch += idx >= k * per_ch ? 1 : 0;
  return ch;
}

// all threads: xs[ch][f][k] = (f16) x[b,f,16ch+k], zero padded.  U 16-byte pieces per thread are fetched before the first is
// converted (branch-free raw buffer loads: one HBM latency per pass, not one per piece; U = 2 NC covers K = 64 at 512 threads and
// K = 16 at 256 in ONE pass -- while the image is staged nothing else runs in the workgroup).  VEC: four consecutive kin are one
// aligned 16-byte piece (x_chunk and K multiples of 4); else element by element.
template <int NC>
struct XStage {
  static constexpr int U = NC >= 2 ? 2 * NC : 4;
};
// the pieces base + u nthreads, u < U, of sample b as fp32 (on = false: every offset out of range, nothing moves)
template <int NC, bool VEC>
__device__ __forceinline__ void stage_x_load(__amdgpu_buffer_rsrc_t rx, const AttnDims& d, int b, int base, int nthreads, bool on,
                                             f32x4 (&v)[XStage<NC>::U]) {
  const int per_ch = d.FP * 4, total = NC * per_ch;
#pragma unroll
  for (int u = 0; u < XStage<NC>::U; ++u) {
    const int idx = base + u * nthreads;
    const int ch = chunk_of<NC>(idx, per_ch), rem = idx - ch * per_ch;
    const int f = rem >> 2, k0 = 16 * ch + 4 * (rem & 3);
    const bool rowok = on && idx < total && f < d.F;
    if constexpr (VEC) {
      v[u] = buf_load4(rx, (rowok && k0 < d.K) ? 4 * x_off(d, b, f, k0) : kOOB);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[u][s] = buf_load1(rx, (rowok && k0 + s < d.K) ? 4 * x_off(d, b, f, k0 + s) : kOOB);
    }
  }
}
template <int NC>
__device__ __forceinline__ void stage_x_write(_Float16* xs, const AttnDims& d, int base, int nthreads, const f32x4 (&v)[XStage<NC>::U]) {
  const int per_ch = d.FP * 4, total = NC * per_ch;
#pragma unroll
  for (int u = 0; u < XStage<NC>::U; ++u) {
    const int idx = base + u * nthreads;
    if (idx < total) {
      const int ch = chunk_of<NC>(idx, per_ch), rem = idx - ch * per_ch;
      const int f = rem >> 2, p = rem & 3;
      *reinterpret_cast<f16x4*>(xs + ch * d.FP * 16 + f * 16 + 4 * (p ^ swz16(f))) = __builtin_convertvector(v[u], f16x4);
    }
  }
}
template <int NC, bool VEC>
__device__ __forceinline__ void stage_x_f16_pass(__amdgpu_buffer_rsrc_t rx, _Float16* xs, const AttnDims& d, int b, int nthreads) {
  const int total = NC * d.FP * 4;
  for (int base = threadIdx.x; base < total; base += XStage<NC>::U * nthreads) {
    f32x4 v[XStage<NC>::U];
    stage_x_load<NC, VEC>(rx, d, b, base, nthreads, true, v);
    stage_x_write<NC>(xs, d, base, nthreads, v);
  }
}
template <int NC>
__device__ __forceinline__ void stage_x_f16(const float* __restrict__ x, _Float16* xs, const AttnDims& d, int b, int nthreads) {
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (long)d.B * d.F * d.K * 4);
  if ((d.xcw & 3) == 0 && (d.K & 3) == 0) {     // (wave-uniform: the two forms as two loops, not a test per piece)
    stage_x_f16_pass<NC, true>(rx, xs, d, b, nthreads);
  } else {
    stage_x_f16_pass<NC, false>(rx, xs, d, b, nthreads);
  }
}

// One dword per 64 bytes of sample b's x rows, into a scrap LDS word by LDS-DMA (no registers, nothing to wait for): the lines
// are in L2 / the Infinity Cache when stage_x_f16 asks for them.  The persistent backward calls it for its NEXT sample while the
// current one is in its last phases; b >= B touches nothing (offsets beyond the descriptor).
template <int NC>
__device__ __forceinline__ void prefetch_x_lines(const float* __restrict__ x, float* scrap, const AttnDims& d, int b, int nthreads) {
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, b < d.B ? (long)d.B * d.F * d.K * 4 : 0);
  const int per_ch = d.FP * 4, total = NC * per_ch;        // 16-byte pieces
  for (int idx = 4 * (int)threadIdx.x; idx < total; idx += 4 * nthreads) {     // (one touch per 64 bytes: rows of 16 floats)
    const int ch = chunk_of<NC>(idx, per_ch), rem = idx - ch * per_ch;
    const int f = rem >> 2, k0 = 16 * ch + 4 * (rem & 3);
    const int off = (f < d.F && k0 < d.K) ? 4 * x_off(d, b, f, k0) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)scrap, 4, off, 0, 0, 0);
  }
}

// weight fragment in the "reduce over kin" role: element s = W[kin = 16c + 4g + s][h][a = lane&15]
// (A operand of (x W)^T products, B operand of x W products)
template <int NC, bool F16>
__device__ __forceinline__ void load_w_kin(const float* __restrict__ W, int h, const AttnDims& d, int lane,
                                           typename Prec<F16>::Op (&w)[NC]) {
  // (raw buffer loads: an element outside [K] x [A] -- or a missing matrix -- is an out-of-range offset that reads 0; as a branch per
  // element these 12 NC loads were ~1000 instructions per wave, paid per SAMPLE by the forward's one-sample workgroups)
  const int a = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(W, W != nullptr ? (long)d.K * d.H * d.A * 4 : 0);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 16 * c + 4 * g + s;
      v[s] = buf_load1(rw, (k < d.K && a < d.A) ? 4 * ((k * d.H + h) * d.A + a) : kOOB);
    }
    w[c] = to_op<F16>(v);
  }
}

// k image of one head (== v): kimg[f][a] = (x Wk)[f][a], written as row fragments from the transposed product
template <int NC, bool F16, typename XS>
__device__ __forceinline__ void project_k(const XS& xsrc, typename Prec<F16>::Elem* kimg, const typename Prec<F16>::Op (&wk)[NC],
                                          int first, int step, int nblk, int lane) {
  // two blocks at a time: a block is one chain (operand reads -> NC dependent products -> convert -> write), and while the k images
  // are projected nothing else runs in the workgroup to cover it
  // (one-chunk shapes only: with NC = 4 the second chain's operands push the K = 64 backward into scratch, 539 -> 559 us)
  int blk = first;
  for (; NC == 1 && blk + step < nblk; blk += 2 * step) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      acc0 = mma<F16>(wk[c], xsrc.row(blk, c, lane), acc0);   // D[a 4g+r][f lane&15]
      acc1 = mma<F16>(wk[c], xsrc.row(blk + step, c, lane), acc1);
    }
    row_write<F16>(kimg, 16 * blk + (lane & 15), lane >> 4, to_op<F16>(acc0));
    row_write<F16>(kimg, 16 * (blk + step) + (lane & 15), lane >> 4, to_op<F16>(acc1));
  }
  for (; blk < nblk; blk += step) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) acc = mma<F16>(wk[c], xsrc.row(blk, c, lane), acc);
    row_write<F16>(kimg, 16 * blk + (lane & 15), lane >> 4, to_op<F16>(acc));
  }
}

// Lane (g,c) of a 16-row block owns 4 consecutive a (a0 = 4g) of row f of one (head, sample)'s [F][A] slab.  The slab is
// a raw buffer of F*A*4 bytes: rows f >= F fall outside it by themselves, columns a >= A are pushed out with kOOB.
struct SlabLane {
  int A, a0;     // row width, first of the lane's four columns
  __device__ __forceinline__ void init(int A_, int a0_) {
    A = A_;
    a0 = a0_;
  }
  __device__ __forceinline__ int off(int f, int s) const { return a0 + s < A ? 4 * (f * A + a0 + s) : kOOB; }
  // Branch-free: BOTH forms are issued, the one that does not apply with an out-of-range offset (it reads zeros and moves no
  // data), and the results are OR-ed.  A branch between the two forms costs far more than four dead load instructions: behind
  // it the compiler's wait-count bookkeeping falls back to vmcnt(0), i.e. whoever consumes an EARLIER prefetch also waits for
  // the loads this call just issued (the backward's per-block prefetch was serialised that way).
  __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t r, int f) const {
    const bool vec = (A & 3) == 0;                         // the four elements are one aligned 16-byte piece
    // (combined as INTEGER vectors: or-ing per element through float <-> int bit casts makes this compiler narrow the
    // 16-byte load to its first dword and use that for all four elements)
    const u32x4 u4 = __builtin_amdgcn_raw_buffer_load_b128(r, vec ? off(f, 0) : kOOB, 0, 0);
    u32x4 u1;
#pragma unroll
    for (int s = 0; s < 4; ++s) u1[s] = __builtin_amdgcn_raw_buffer_load_b32(r, vec ? kOOB : off(f, s), 0, 0);
    return __builtin_bit_cast(f32x4, u4 | u1);
  }
  // A16 = true: the row width is known to be 16 (one aligned 16-byte piece per lane, no second form)
  template <bool A16>
  __device__ __forceinline__ f32x4 load_t(__amdgpu_buffer_rsrc_t r, int f) const {
    if constexpr (A16) {
      return buf_load4(r, 4 * (f * 16 + a0));
    } else {
      return load(r, f);
    }
  }
  template <bool A16>
  __device__ __forceinline__ void store_t(__amdgpu_buffer_rsrc_t r, int f, const f32x4& v) const {
    if constexpr (A16) {
      buf_store4(r, 4 * (f * 16 + a0), v);
    } else {
      store(r, f, v);
    }
  }
  __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int f, const f32x4& v) const {
    if ((A & 3) == 0) {
      buf_store4(r, off(f, 0), v);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) buf_store1(r, off(f, s), v[s]);
    }
  }
};

// Weight fragments of the backward.  f16 mode: an LDS table of 16x16 tiles, tile (m, hh, cc)[kin][a] = W_m[16cc+kin][hh][a]
// (m = 0 q, 1 k, 2 res), built once per workgroup; row reads give the "reduce over a" fragments, transposing reads the
// "reduce over kin" ones.  f32 mode: the table would not fit beside the fp32 images, fragments come from global / L1.
template <bool F16>
struct WTab;
template <>
struct WTab<true> {
  const _Float16* tab;
  int nw, NC;
  __device__ __forceinline__ f16x4 kin(int m, int hh, int cc, int lane) const {          // W[16cc + 4g + s][hh][a = lane&15]
    return tr_read<true>(tab + ((m * nw + hh) * NC + cc) * 256, 0, lane);
  }
  __device__ __forceinline__ f16x4 arole(int m, int hh, int cc, int lane) const {        // W[16cc + (lane&15)][hh][a = 4g + s]
    return row_read<true>(tab + ((m * nw + hh) * NC + cc) * 256, lane & 15, lane >> 4);
  }
};
template <>
struct WTab<false> {
  __amdgpu_buffer_rsrc_t rw[3];
  int H, A, K;
  __device__ __forceinline__ void init(const float* Wq, const float* Wk, const float* Wr, const AttnDims& d) {
    const long bytes = (long)d.K * d.H * d.A * 4;
    rw[0] = make_rsrc(Wq, bytes);
    rw[1] = make_rsrc(Wk, bytes);
    rw[2] = make_rsrc(Wr != nullptr ? Wr : Wq, Wr != nullptr ? bytes : 0);
    H = d.H; A = d.A; K = d.K;
  }
  __device__ __forceinline__ f32x4 kin(int m, int hh, int cc, int lane) const {
    const int a = lane & 15, k0 = 16 * cc + 4 * (lane >> 4);
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = buf_load1(rw[m], (a < A && k0 + s < K) ? 4 * (((k0 + s) * H + hh) * A + a) : kOOB);
    return v;
  }
  __device__ __forceinline__ f32x4 arole(int m, int hh, int cc, int lane) const {
    const int kin_ = 16 * cc + (lane & 15), a0 = 4 * (lane >> 4);
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = buf_load1(rw[m], (kin_ < K && a0 + s < A) ? 4 * ((kin_ * H + hh) * A + a0 + s) : kOOB);
    return v;
  }
};

// workgroup barrier that orders LDS traffic only: global stores and prefetches stay in flight (a __syncthreads() also
// drains vmcnt).  Inline asm: the waitcnt pass does not look inside, the "memory" clobber pins the compiler's own order.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ================================================================================================= forward
// grid = B, block = 64 H.  y[h,b,f,a] = fuse_relu ? relu(res + ln) : ln ; res_out (optional, !fuse_relu) = x Wr
// KR > 0 (f16 mode, at most KR key tiles): the head's k fragments live in REGISTERS, in both orientations -- the projection's
// transposed product D[a][f] IS the row fragment of a k tile, the product with the operands swapped is its column fragment
// (as the backward does for q) -- 4 KR registers instead of an LDS image that every query block re-reads tile by tile: two of a
// tile's ~21 instructions and both of its LDS waits are gone (the kernel has the registers: 44-74 of 512 / waves).
template <int NC, bool F16, bool A16, int KR = 0>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                        const float* __restrict__ Wk, const float* __restrict__ Wr,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, float* __restrict__ res_out,
                                                        float* __restrict__ av_out, float* __restrict__ rstd_out, AttnDims d,
                                                        float scale, float eps, int fuse_relu) {
  typedef typename Prec<F16>::Elem Elem;
  typedef typename Prec<F16>::Op Op;
  constexpr int RS = Prec<F16>::RS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keep it scalar
  const int c = lane & 15, g = lane >> 4;
  static_assert(KR == 0 || F16, "register k fragments are an f16-mode form");
  // (several chunks: the weight fragments are requested first, their round trip to L2 runs under the staging of x -- K = 64 forward
  // 187.4 -> 182.5 us; with one chunk the order made the kernel 1 % slower, 152.4 -> 154.2 us, and stays as it was)
  Op wq[NC], wk[NC], wr[NC];
  if constexpr (NC >= 2) {
    load_w_kin<NC, F16>(Wq, h, d, lane, wq);
    load_w_kin<NC, F16>(Wk, h, d, lane, wk);
    load_w_kin<NC, F16>(Wr, h, d, lane, wr);
  }
  Elem* kimg = nullptr;
  XSrc<F16, F16> xsrc;
  if constexpr (F16) {
    _Float16* xs = reinterpret_cast<_Float16*>(smem_raw);
    stage_x_f16<NC>(x, xs, d, b, blockDim.x);
    xsrc.xs = xs;
    xsrc.FP = d.FP;
    if constexpr (KR == 0) kimg = xs + d.NC * d.FP * 16 + h * d.FP * RS;
    __syncthreads();
  } else {
    xsrc.init(x, d);
    xsrc.set_sample(d, b);
    kimg = reinterpret_cast<float*>(smem_raw) + h * d.FP * RS;
  }
  if constexpr (NC < 2) {
    load_w_kin<NC, F16>(Wq, h, d, lane, wq);
    load_w_kin<NC, F16>(Wk, h, d, lane, wk);
    load_w_kin<NC, F16>(Wr, h, d, lane, wr);
  }
  const bool use_ln = gamma != nullptr;
  SlabLane sl;
  sl.init(d.A, 4 * g);
  const float inv_a = 1.0f / (float)d.A;
  f32x4 gam = {0.f, 0.f, 0.f, 0.f}, bet = {0.f, 0.f, 0.f, 0.f};
  bool aval[4];     // (A16: all true at compile time, the masks fold away)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    aval[s] = A16 || 4 * g + s < d.A;
    if (use_ln && aval[s]) {
      gam[s] = gamma[4 * g + s];
      bet[s] = beta[4 * g + s];
    }
  }
  Op kRow[KR > 0 ? KR : 1], kCol[KR > 0 ? KR : 1];   // k[key c][a 4g..], k[key 4g..][a c] of tile t
  if constexpr (KR > 0) {
#pragma unroll
    for (int t = 0; t < KR; ++t) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (t < d.nblk) {
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          const Op xr = xsrc.row(t, cc, lane);
          a0 = mma<F16>(wk[cc], xr, a0);     // D[a 4g+r][key c]
          a1 = mma<F16>(xr, wk[cc], a1);     // D[key 4g+r][a c]
        }
      }
      kRow[t] = to_op<F16>(a0);
      kCol[t] = to_op<F16>(a1);
    }
  } else {
    project_k<NC, F16>(xsrc, kimg, wk, 0, 1, d.nblk, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  const float qs = -scale * 1.4426950408889634f;
  const long slab = ((long)h * d.B + b) * d.F * d.A;   // this (head, sample)'s [F][A] rows
  const long slab_bytes = (long)d.F * d.A * 4;
  const __amdgpu_buffer_rsrc_t r_y = make_rsrc(y + slab, slab_bytes);
  // the rows kept for the backward are the NORMALISED ones, with their 1/sigma beside them (rstd_out): its LayerNorm gradient starts
  // from them instead of deriving mean and variance of every row again
  const bool keep_stats = use_ln && rstd_out != nullptr;
  const __amdgpu_buffer_rsrc_t r_av = make_rsrc(keep_stats ? av_out + slab : y, keep_stats ? slab_bytes : 0);
  const __amdgpu_buffer_rsrc_t r_res = make_rsrc(res_out != nullptr ? res_out + slab : y, res_out != nullptr ? slab_bytes : 0);
  const __amdgpu_buffer_rsrc_t r_rs = make_rsrc(keep_stats ? rstd_out + ((long)h * d.B + b) * d.F : y, keep_stats ? (long)d.F * 4 : 0);
  for (int i = 0; i < d.nblk; ++i) {
    Op xr[NC];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) xr[cc] = xsrc.row(i, cc, lane);
    f32x4 qT = {0.f, 0.f, 0.f, 0.f}, resT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) qT = mma<F16>(wq[cc], xr[cc], qT);          // [a 4g+r][query c]
    if (Wr != nullptr) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) resT = mma<F16>(wr[cc], xr[cc], resT);
    }
    const Op qn = to_op<F16>(qT * qs);
    f32x4 avT = {0.f, 0.f, 0.f, 0.f};
    // straight-line tiles with a scalar exit test each (F <= 512: at most 32): the tile index is a compile-time constant, so the
    // two LDS reads of a tile use immediate offsets -- as a counted loop every tile paid two vector adds for its addresses, two of
    // the ~21 vector-issue slots of a tile in a loop that is bound by them
    {
      auto fwd_tile = [&](auto Tc) __attribute__((always_inline)) {
        constexpr int t = decltype(Tc)::value;
        if constexpr (KR > 0 && t >= KR) return;      // (never reached: nblk <= KR)
        Op kA, kT;
        if constexpr (KR > 0) {
          kA = kRow[t < KR ? t : 0];
          kT = kCol[t < KR ? t : 0];
        } else {
          kA = row_read<F16>(kimg, 16 * t + c, g);
          kT = tr_read<F16>(kimg, 16 * t, lane);
        }
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = mma<F16>(kA, qn, sc);                                                  // [key 4g+r][query c], times -log2e*scale
        f32x4 sg;
#pragma unroll
        for (int r = 0; r < 4; ++r) sg[r] = sigmoid_from_neg_log2(sc[r]);
        avT = mma<F16>(kT, to_op<F16>(sg), avT);                                    // [a 4g+r][query c]
      };
      int nb_s = d.nblk;
      asm volatile("" : "+s"(nb_s));
#define FIL_FWD_TILE(T)                                  \
  if (T >= nb_s) goto fwd_tiles_done;                    \
  fwd_tile(std::integral_constant<int, T>{});
      FIL_FWD_TILE(0) FIL_FWD_TILE(1) FIL_FWD_TILE(2) FIL_FWD_TILE(3) FIL_FWD_TILE(4) FIL_FWD_TILE(5) FIL_FWD_TILE(6) FIL_FWD_TILE(7)
      FIL_FWD_TILE(8) FIL_FWD_TILE(9) FIL_FWD_TILE(10) FIL_FWD_TILE(11) FIL_FWD_TILE(12) FIL_FWD_TILE(13) FIL_FWD_TILE(14) FIL_FWD_TILE(15)
      FIL_FWD_TILE(16) FIL_FWD_TILE(17) FIL_FWD_TILE(18) FIL_FWD_TILE(19) FIL_FWD_TILE(20) FIL_FWD_TILE(21) FIL_FWD_TILE(22) FIL_FWD_TILE(23)
      FIL_FWD_TILE(24) FIL_FWD_TILE(25) FIL_FWD_TILE(26) FIL_FWD_TILE(27) FIL_FWD_TILE(28) FIL_FWD_TILE(29) FIL_FWD_TILE(30) FIL_FWD_TILE(31)
#undef FIL_FWD_TILE
    fwd_tiles_done:;
    }
    // lane (g,c): av[query 16i+c][a 4g..4g+3]
    const int f = 16 * i + c;
    f32x4 ln = avT;
    if (use_ln) {
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) sum += aval[s] ? avT[s] : 0.f;
      const float mu = groups_sum(sum) * inv_a;
      f32x4 dv;
      float sq = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        dv[s] = aval[s] ? avT[s] - mu : 0.f;
        sq = fmaf(dv[s], dv[s], sq);
      }
      const float rstd = __builtin_amdgcn_rsqf(groups_sum(sq) * inv_a + eps);   // v_rsq_f32, 1 ulp
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        dv[s] *= rstd;
        ln[s] = dv[s] * gam[s] + bet[s];
      }
      sl.template store_t<A16>(r_av, f, dv);                  // zero-size descriptors drop the stores when nothing is kept
      buf_store1(r_rs, 4 * f + (g == 0 ? 0 : kOOB), rstd);
    }
    if (fuse_relu) {
      f32x4 o;
#pragma unroll
      for (int s = 0; s < 4; ++s) o[s] = fmaxf(resT[s] + ln[s], 0.f);
      sl.template store_t<A16>(r_y, f, o);
    } else {
      sl.template store_t<A16>(r_y, f, ln);
      sl.template store_t<A16>(r_res, f, resT);
    }
  }
}

// ================================================================================================= backward
// Persistent: grid = G workgroups of 64 H threads (wave = head), workgroup w takes samples w, w+G, ...   NB = compile-time
// bound on nblk (the dk accumulators of all key tiles live in registers).
// LDS: [x image (f16 mode, K <= 32)] [k images: H] [tiles: H x 6] [weight table: 3 x H x NC tiles, built once]
//   tiles of wave h: 0,1 = dP ping-pong (0 also turns dav);  2+2p / 3+2p = dq / dres of the step with parity p -- they turn
//   the wave's own fragments AND are the hand-off to the wave that owns a 16-column chunk of dx (sum over the heads in
//   MFMA accumulators: fixed order, no dx tile in LDS).  The dk part of dx is added in a second visit by the same lanes.
// Outputs: dx, per-workgroup partials of dWq/dWk/dWr [G][3][K][H][A] and of dgamma/dbeta [G*H][2][16].
#ifndef FIL_ATTN_TILE_GROUP
#define FIL_ATTN_TILE_GROUP 1
#endif
#ifndef FIL_ATTN_EXIT_TEST
#define FIL_ATTN_EXIT_TEST(J) (J >= nb_s)     // (experiments: -D'FIL_ATTN_EXIT_TEST(J)=false' = no exit test, nblk == NB only)
#endif
#ifndef FIL_ATTN_XL_MAXNC
#define FIL_ATTN_XL_MAXNC 4
#endif
#ifndef FIL_ATTN_BWD_WPE
#define FIL_ATTN_BWD_WPE(NC, F16) 2
#endif
// WPH = waves per head.  1: a head is one wave.  2: two waves per head take alternate QUERY blocks (block i = 2 step + sub):
// nothing is computed twice (each wave runs the whole prologue + all key tiles of its own blocks and keeps its own partial dk /
// dW / dgamma sums, merged after the loop), and a workgroup has twice the waves -- for the shapes whose LDS footprint lets only
// one workgroup onto a CU (K = 64 layers of a stack, the f32 mode at large F) that is the second wave per SIMD.
template <int NC, bool F16, int NB, int WPH, bool DXL, bool A16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(FIL_ATTN_BWD_WPE(NC, F16)))) void attn_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wr,
    const float* __restrict__ gamma, const float* __restrict__ dy, const float* __restrict__ dres_in,
    const float* __restrict__ y_s, const float* __restrict__ av_s, const float* __restrict__ rstd_s, float* __restrict__ dx,
    float* __restrict__ wpart, float* __restrict__ gb_part, AttnDims d, float scale, float eps, int fuse_relu, long long* __restrict__ stamps) {
  typedef typename Prec<F16>::Elem Elem;
  typedef typename Prec<F16>::Op Op;
  constexpr int RS = Prec<F16>::RS;
  constexpr int TS = 16 * RS;
  // diagnostic build only (-DFIL_ATTN_STAMPS): shader-clock time per phase and wave, written to a buffer of its own
#ifdef FIL_ATTN_STAMPS
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = __builtin_amdgcn_s_memtime();
#define FIL_STAMP_AT(P) { const long long t_now = __builtin_amdgcn_s_memtime(); ph[P] += t_now - t_last; t_last = t_now; }
#ifdef FIL_ATTN_STAMPS_POST      // breakdown of the per-sample phases instead: the whole block loop in slot 1, FIL_STAMP_POST(k) in 2..6
#define FIL_STAMP(P) FIL_STAMP_AT(((P) >= 1 && (P) <= 5) ? 1 : (P))
#define FIL_STAMP_POST(P) FIL_STAMP_AT(P)
#else
#define FIL_STAMP(P) FIL_STAMP_AT(P)
#define FIL_STAMP_POST(P)
#endif
#else
#define FIL_STAMP(P)
#define FIL_STAMP_POST(P)
#endif               // elements per 16-row tile
  constexpr bool XL = F16 && NC <= FIL_ATTN_XL_MAXNC;   // x image in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  static_assert(WPH == 1 || WPH == 2, "one or two waves per head");
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;   // == H * WPH
  const int NH = d.H;
  // this wave's head; its query blocks are i = WPH * step + sub.  Two waves per head: wave w is (head w mod H, sub w / H), so that the
  // two waves of a head share a SIMD (waves of a workgroup go to the SIMDs round robin, H = 4): when only the sub-0 waves have work
  // -- the single-handed last step of an odd block count, each half of the dk merge -- every SIMD has ONE busy wave instead of two
  // SIMDs having two and two having none (head-major numbering, w = 2 h + sub, did that)
  const int sub = (WPH > 1 && w >= NH) ? 1 : 0, h = w - sub * NH;
  auto wave_of = [&](const int hh, const int s_) __attribute__((always_inline)) { return WPH > 1 ? s_ * NH + hh : hh; };
  const int c = lane & 15, g = lane >> 4;
  unsigned char* sp = smem_raw;
  XSrc<F16, XL> xsrc;
  _Float16* xs16 = nullptr;
  if constexpr (XL) {
    xs16 = reinterpret_cast<_Float16*>(sp);
    sp += (size_t)d.NC * d.FP * 16 * sizeof(_Float16);
    xsrc.xs = xs16;
    xsrc.FP = d.FP;
  } else {
    xsrc.init(x, d);
  }
  const int KIS = (d.FP + 16) * RS;           // k image of one head: FP rows + one zero tile (the pipeline reads one tile ahead)
  Elem* kimg0 = reinterpret_cast<Elem*>(sp);
  Elem* kimg = kimg0 + h * KIS;               // shared by the head's waves
  sp += (size_t)NH * KIS * sizeof(Elem);
  Elem* tiles0 = reinterpret_cast<Elem*>(sp);
  Elem* tiles = tiles0 + w * 6 * TS;          // per wave
  sp += (size_t)nw * 6 * TS * sizeof(Elem);
  const bool use_ln = gamma != nullptr, has_res = Wr != nullptr;
  SlabLane sl;
  sl.init(d.A, 4 * g);
  const float inv_a = 1.0f / (float)d.A;
  const __amdgpu_buffer_rsrc_t r_dx = make_rsrc(dx, (long)d.B * d.F * d.K * 4);
  WTab<F16> wt;
  if constexpr (F16) {
    _Float16* wtab = reinterpret_cast<_Float16*>(sp);     // tile (m, hh, cc) at ((m*nw + hh)*NC + cc)*256
    wt.tab = wtab;
    wt.nw = NH;
    wt.NC = NC;
    const float* Wm[3] = {Wq, Wk, Wr};
    const int ntile = 3 * NH * NC;
    for (int idx = threadIdx.x; idx < ntile * 64; idx += blockDim.x) {
      const int tl = idx >> 6, kin_l = (idx >> 2) & 15, a4 = idx & 3;
      const int m = tl / (NH * NC), rem = tl - m * NH * NC, hh = rem / NC, cc = rem - hh * NC;
      const int kin = 16 * cc + kin_l;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (Wm[m] != nullptr && kin < d.K) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (4 * a4 + s < d.A) v[s] = Wm[m][((long)kin * d.H + hh) * d.A + 4 * a4 + s];
      }
      row_write<true>(wtab + tl * 256, kin_l, a4, to_op<true>(v));
    }
  } else {
    wt.init(Wq, Wk, Wr, d);
  }
  // XFIX (K = 64, two waves per head, four heads: eight dx jobs per step on eight waves): wave w always takes job w -- block
  // 2 st + w / 4, chunk w % 4 -- so the eight weight fragments of its chunk stay in registers instead of being read from the LDS
  // table in every step (rotating the owner spreads nothing when every wave has exactly one job)
  // (only where the sixteen registers exist: the attention_dim == 16, NB = 13 instantiation -- the c5 stack's -- has them; the general
  // column-masked form and NB = 32 spill 17-50 more)
  constexpr bool XFIX = F16 && DXL && WPH == 2 && NC == 4 && A16 && NB <= 13;
  const bool xfix = XFIX && NH == 4;
  Op xwq[4], xwr[4];
  if constexpr (XFIX) {
    lds_barrier();   // the weight table is complete
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      xwq[hh] = wt.arole(0, xfix ? hh : 0, w & 3, lane);
      xwr[hh] = wt.arole(2, xfix ? hh : 0, w & 3, lane);
    }
  }
  // dx of the sample being processed, fp32 [NC][FP][16] (chunk-major like the head-major global layout), when the footprint
  // allows (dx_lds): the heads' dq / dres parts are summed into it block by block, the dk part is added after the block loop,
  // and it leaves once, as whole 1-KB rows.  Without it the dq part goes to global memory and is read back for the dk part
  // ("second visit": +2 x |dx| of HBM traffic per launch -- 420 MB at K = 64, F = 200, B = 4096 -- in 64-byte pieces).
  // (DXL: a compile-time choice -- this kernel sits at its register limit, a run-time branch costs it 100 more spilled registers)
  float* dxs = nullptr;
  if constexpr (DXL) dxs = reinterpret_cast<float*>(sp + (F16 ? (size_t)3 * NH * NC * 256 * sizeof(_Float16) : 0));
  row_write<F16>(kimg, d.FP + c, g, to_op<F16>(f32x4{0.f, 0.f, 0.f, 0.f}));   // the zero tile (never written again)
  __shared__ __attribute__((aligned(16))) float gamma_s[16];      // re-read per block: 4 registers less than keeping it
  __shared__ float x_scrap[64];                                    // where prefetch_x_lines drops its dwords (never read)
  if (threadIdx.x < 16) gamma_s[threadIdx.x] = (use_ln && (int)threadIdx.x < d.A) ? gamma[threadIdx.x] : 0.f;
  bool aval[4];     // (A16: all true at compile time)
#pragma unroll
  for (int s = 0; s < 4; ++s) aval[s] = A16 || 4 * g + s < d.A;
  f32x4 dWq[NC], dWk[NC], dWr[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) dWq[cc] = dWk[cc] = dWr[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dgam = {0.f, 0.f, 0.f, 0.f}, dbet = {0.f, 0.f, 0.f, 0.f};
  const float qs = -scale * 1.4426950408889634f;
  // "reduce over kin" weight fragments of this head: the f32 mode keeps them in registers (they would come from global
  // memory on every use), the f16 mode re-reads its LDS table (registers are what limits its occupancy)
  Op wq_r[NC], wk_r[NC];
  if constexpr (!F16) {
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      wq_r[cc] = wt.kin(0, h, cc, lane);
      wk_r[cc] = wt.kin(1, h, cc, lane);
    }
  }

  // f16 mode, when one pass of the staging loop covers the whole image (every BASELINE shape): the NEXT sample's x is requested
  // right after this sample's last use of the image (dWk) and converted into it at the top of the next sample -- the round trip
  // (12 % of the K = 64 kernel as a serial phase) runs under the dk part of dx, and ahead of this sample's dx stores in the queue.
  // (The loads are issued for every sample, out of range when they do not apply: a conditionally written xpre would be live
  // through the block loop.)
  const bool x_early = XL && (d.xcw & 3) == 0 && (d.K & 3) == 0 && NC * d.FP * 4 <= XStage<NC>::U * (int)blockDim.x;
  f32x4 xpre[XStage<NC>::U];
  for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
    FIL_STAMP(7)
    lds_barrier();   // weight table built / the previous sample's k (dk) images and x image are no longer read
    if constexpr (XL) {
      if (x_early && b != (int)blockIdx.x) {
        stage_x_write<NC>(xs16, d, threadIdx.x, blockDim.x, xpre);   // requested after the previous sample's dWk (below)
      } else {
        stage_x_f16<NC>(x, xs16, d, b, blockDim.x);
      }
      lds_barrier();
      FIL_STAMP_POST(6)    // x staged
    } else {
      xsrc.set_sample(d, b);
    }
    // this (head, sample)'s [F][A] rows as raw buffers; a tensor that is not used gets a zero-size descriptor (reads 0)
    const long slab = ((long)h * d.B + b) * d.F * d.A;
    const long slab_bytes = (long)d.F * d.A * 4;
    const bool use_dr = !fuse_relu && has_res && dres_in != nullptr;
    const __amdgpu_buffer_rsrc_t r_dy = make_rsrc(dy + slab, slab_bytes);
    const __amdgpu_buffer_rsrc_t r_ys = make_rsrc(fuse_relu ? y_s + slab : dy, fuse_relu ? slab_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_avs = make_rsrc(use_ln ? av_s + slab : dy, use_ln ? slab_bytes : 0);
    const __amdgpu_buffer_rsrc_t r_dr = make_rsrc(use_dr ? dres_in + slab : dy, use_dr ? slab_bytes : 0);
    // av_s: the NORMALISED LayerNorm input rows, rstd_s their 1/sigma, as the forward leaves them (the statistics are not derived again)
    const __amdgpu_buffer_rsrc_t r_rs = make_rsrc(use_ln ? rstd_s + ((long)h * d.B + b) * d.F : dy, use_ln ? (long)d.F * 4 : 0);

    // block inputs are fetched one query block ahead (they come from HBM)
    // (dres_in of the unfused mode is read at use: one more prefetched tensor would cost the fused mode a wave per SIMD)
    f32x4 n_dy = sl.template load_t<A16>(r_dy, 16 * sub + c), n_y = sl.template load_t<A16>(r_ys, 16 * sub + c), n_av = sl.template load_t<A16>(r_avs, 16 * sub + c);
    float n_rs = buf_load1(r_rs, 4 * (16 * sub + c));

    {
      Op wk[NC];
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) wk[cc] = F16 ? wt.kin(1, h, cc, lane) : wk_r[cc];
      project_k<NC, F16>(xsrc, kimg, wk, sub, WPH, d.nblk, lane);   // the head's waves share the blocks
    }
    if constexpr (WPH > 1) {
      lds_barrier();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    f32x4 dk[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) dk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the first block's inputs, requested before the k projection: waited for once, here, so that the block loop is entered
    // with nothing pending -- see the end of phase E)
    settle4(n_dy);
    settle4(n_y);
    settle4(n_av);
    asm volatile("" : "+v"(n_rs));
    FIL_STAMP(0)
    // ---- The block loop.  A query block i goes through four phases:
    //   P(i)  LayerNorm / ReLU backward of the block's rows, q_i in both orientations; requests the inputs of the wave's next block
    //   T(i)  the score tiles (all key tiles against the block; dk accumulates in registers, dq_i in dqT)
    //   E(i)  dq_i / dres_i into the wave's hand-over tiles, dW += x_i^T (dq_i | dres_i)
    //   X     dx_i = sum over the heads of dq Wq^T + dres Wr^T, from the hand-over tiles of all heads, behind a workgroup barrier
    // With two waves per head (WPH = 2) step `st` runs blocks 2 st and 2 st + 1 side by side.
    // (Tried in round 5 and dropped, DESIGN.md section 4.4: the next block's P cut into pieces between the current block's tiles; the
    // two waves of a head in alternating phases ("slots": T beside E + X + P); the odd last block's key tiles split between the
    // head's two waves -- all slower or equal: the SIMDs are issue-bound in every phase, so moving instructions between phases or
    // waves buys nothing and every extra select, test or barrier costs.)
    Op qn, qc, dav_r, dav_c, dr_r;     // P -> T, E
    f32x4 dqT = {0.f, 0.f, 0.f, 0.f};  // T -> E
    auto phase_p = [&](const int i) __attribute__((always_inline)) {
      f32x4 dz = n_dy, dr = {0.f, 0.f, 0.f, 0.f};
      if (use_dr) dr = sl.template load_t<A16>(r_dr, 16 * i + c);
      const f32x4 yv = n_y, avv = n_av;
      const float rsv = n_rs;
      {
        const int fn = 16 * (i + WPH) + c;  // past the last block every lane is out of range and reads zeros
        n_dy = sl.template load_t<A16>(r_dy, fn);
        n_y = sl.template load_t<A16>(r_ys, fn);
        n_av = sl.template load_t<A16>(r_avs, fn);
        n_rs = buf_load1(r_rs, 4 * fn);
      }
      // ---- LayerNorm / ReLU backward of query block i: lane (g,c) owns row f = 16i+c, a = 4g..4g+3
      if (fuse_relu) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (!(yv[s] > 0.f)) dz[s] = 0.f;
        dr = dz;
      }
      f32x4 dav = dz;
      if (use_ln) {
        f32x4 xh;
#pragma unroll
        for (int s = 0; s < 4; ++s) xh[s] = aval[s] ? avv[s] : 0.f;
        const float rstd = rsv;
        float s1 = 0.f, s2 = 0.f;
        f32x4 dxh;
        const f32x4 gam = *reinterpret_cast<const f32x4*>(gamma_s + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dgam[s] = fmaf(dz[s], xh[s], dgam[s]);
          dbet[s] += dz[s];
          dxh[s] = dz[s] * gam[s];
          s1 += dxh[s];
          s2 = fmaf(dxh[s], xh[s], s2);
        }
        const float m1 = groups_sum(s1) * inv_a, m2 = groups_sum(s2) * inv_a;
#pragma unroll
        for (int s = 0; s < 4; ++s) dav[s] = aval[s] ? rstd * (dxh[s] - m1 - xh[s] * m2) : 0.f;
      }
      dav_r = to_op<F16>(dav);     // row fragments: [query c][a 4g+s]
      dr_r = to_op<F16>(dr);
      row_write<F16>(tiles, c, g, dav_r);                       // tile 0 (free: the wave's last tile phase is over)
      dav_c = tr_read<F16>(tiles, 0, lane);                     // column fragment [query 4g+s][a c]
      // ---- q_i in both orientations
      f32x4 qT = {0.f, 0.f, 0.f, 0.f}, qD = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const Op xr = xsrc.row(i, cc, lane);
        const Op wq = F16 ? wt.kin(0, h, cc, lane) : wq_r[cc];
        qT = mma<F16>(wq, xr, qT);   // [a 4g+r][query c] -> row fragment of q_i
        qD = mma<F16>(xr, wq, qD);   // [query 4g+r][a c] -> column fragment of q_i
      }
      qn = to_op<F16>(qT * qs);     // scores come out as -log2(e) * scale * q.k
      qc = to_op<F16>(qD * scale);  // dk += dP^T (scale q)
    };
    auto phase_t = [&]() __attribute__((always_inline)) {
      dqT = f32x4{0.f, 0.f, 0.f, 0.f};
      // ---- the score tiles, software-pipelined over three tiles: [LDS reads of tile j+1] [sigmoid + dk of tile j]
      // [S, dS of tile j+1] [dq of tile j-1].  The k image carries one zero tile behind the last key block, so the reads
      // and products of tile j+1 need no guard; leaving the unrolled loop with `break` keeps the exit test scalar.
      Op kB_n = row_read<F16>(kimg, c, g);                    // k[key c][a 4g+s]
      Op kT_n = tr_read<F16>(kimg, 0, lane);                  // k[key 4g+s][a c]
      f32x4 sc_n = mma<F16>(qn, kB_n, f32x4{0.f, 0.f, 0.f, 0.f});        // [query 4g+r][key c]
      f32x4 ds_n = mma<F16>(dav_r, kB_n, f32x4{0.f, 0.f, 0.f, 0.f});     // dS = dav k^T
      Op kT_prev = kT_n;
      // one tile; J is a compile-time index (the dk accumulators must stay in registers)
      auto tile_step = [&](auto Jc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value;
        const f32x4 sc = sc_n, ds = ds_n;
        const Op kT_j = kT_n;
        Op dpT = kT_n;
        if constexpr (j > 0) dpT = tr_read<F16>(tiles + ((j - 1) & 1) * TS, 0, lane);   // dP[query c][key 4g+s] of tile j-1
        kB_n = row_read<F16>(kimg, 16 * (j + 1) + c, g);
        kT_n = tr_read<F16>(kimg, 16 * (j + 1), lane);
        f32x4 sg, dp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(sc[r]);        // exp(-score)
          sg[r] = __builtin_amdgcn_rcpf(1.0f + e);
          dp[r] = ds[r] * (e * sg[r] * sg[r]);                  // dS S (1-S), 1-S = e S; the 1/sqrt(A) factor rides on qc and dq
        }
        const Op dp_o = to_op<F16>(dp), sg_o = to_op<F16>(sg);
        sc_n = mma<F16>(qn, kB_n, f32x4{0.f, 0.f, 0.f, 0.f});
        ds_n = mma<F16>(dav_r, kB_n, f32x4{0.f, 0.f, 0.f, 0.f});
        if constexpr (j > 0) dqT = mma<F16>(kT_prev, dpT, dqT);   // dq^T[a 4g+r][query c] += k^T dP^T
        row_write<F16>(tiles + (j & 1) * TS, c, g, dp_o);     // [key c][query 4g..4g+3]
        dk[j] = mma<F16>(qc, dp_o, dk[j]);                    // dk^T[a 4g+r][key c] += (scale q)^T dP
        dk[j] = mma<F16>(dav_c, sg_o, dk[j]);                 //                     += dav^T S      (V == K)
        kT_prev = kT_j;
        // scheduling fence every FIL_ATTN_TILE_GROUP tiles: inside a group the scheduler may interleave the tiles' chains
        // (instruction-level parallelism for a kernel that runs at two waves per SIMD), across groups it may not (registers)
        if constexpr (j % FIL_ATTN_TILE_GROUP == FIL_ATTN_TILE_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
      };
      // straight-line tiles with a scalar exit test each (a `break` in an unrolled loop re-rolls it and sends dk to scratch;
      // the count is laundered through an empty asm so that the NB exit conditions are not hoisted out of the block loop
      // as NB lane masks, which then live in spilled SGPRs)
      int nb_s = d.nblk;
      asm volatile("" : "+s"(nb_s));
#define FIL_TILE(J)                                             \
  if constexpr (J < NB) {                                       \
    if (FIL_ATTN_EXIT_TEST(J)) goto tiles_done;                 \
    tile_step(std::integral_constant<int, J>{});                \
  }
      FIL_TILE(0) FIL_TILE(1) FIL_TILE(2) FIL_TILE(3) FIL_TILE(4) FIL_TILE(5) FIL_TILE(6) FIL_TILE(7)
      FIL_TILE(8) FIL_TILE(9) FIL_TILE(10) FIL_TILE(11) FIL_TILE(12) FIL_TILE(13) FIL_TILE(14) FIL_TILE(15)
      FIL_TILE(16) FIL_TILE(17) FIL_TILE(18) FIL_TILE(19) FIL_TILE(20) FIL_TILE(21) FIL_TILE(22) FIL_TILE(23)
      FIL_TILE(24) FIL_TILE(25) FIL_TILE(26) FIL_TILE(27) FIL_TILE(28) FIL_TILE(29) FIL_TILE(30) FIL_TILE(31)
#undef FIL_TILE
    tiles_done:
      {
        const Op dpT = tr_read<F16>(tiles + ((d.nblk - 1) & 1) * TS, 0, lane);
        dqT = mma<F16>(kT_prev, dpT, dqT);
      }
    };
    auto phase_e = [&](const int i, const int par) __attribute__((always_inline)) {
      const Op dq_r = to_op<F16>(dqT * scale);                // row fragment of dq_i
      Elem* t_dq = tiles + (2 + 2 * par) * TS;
      row_write<F16>(t_dq, c, g, dq_r);
      if (has_res) row_write<F16>(t_dq + TS, c, g, dr_r);
      const Op dq_c = tr_read<F16>(t_dq, 0, lane);
      Op dr_c = dr_r;
      if (has_res) dr_c = tr_read<F16>(t_dq + TS, 0, lane);
      // ---- dW += x_i^T d*
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        const Op xc = xsrc.col(i, cc, lane);                  // x[query 4g+s][kin c]
        dWq[cc] = mma<F16>(xc, dq_c, dWq[cc]);                // [kin 4g+r][a c]
        if (has_res) dWr[cc] = mma<F16>(xc, dr_c, dWr[cc]);
      }
    };
    // X of step st: job q = (qs', cc) -- block WPH st + qs', chunk cc of its dx -- belongs to wave (q + st) mod nw (rotw = st mod nw,
    // kept by the loop): the owner rotates with the step, so the work -- and the arrival skew it causes at the next barrier -- is
    // spread over the waves.
    auto phase_x = [&](const int st, const int rotw) __attribute__((always_inline)) {
      const int par = st & 1;
      if constexpr (XFIX) {
        if (xfix) {
          const int qs_ = w >> 2, cc = w & 3, bi = st * WPH + qs_;
          if (bi < d.nblk) {
            f32x4 px = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
              const Elem* th = tiles0 + (wave_of(hh, qs_) * 6 + 2 + 2 * par) * TS;
              px = mma<F16>(xwq[hh], row_read<F16>(th, c, g), px);
              if (has_res) px = mma<F16>(xwr[hh], row_read<F16>(th + TS, c, g), px);
            }
            *reinterpret_cast<f32x4*>(dxs + (cc * d.FP + 16 * bi + c) * 16 + 4 * g) = px;
          }
          return;
        }
      }
      for (int q = w >= rotw ? w - rotw : w - rotw + nw; q < WPH * NC; q += nw) {
        const int qs_ = q / NC, cc = q - qs_ * NC, bi = st * WPH + qs_;
        if (bi < d.nblk) {
          // DXL: the TRANSPOSED tile, D[kin 4g+r][query c] = W d^T -- a lane then holds four consecutive kin of one row, one 16-byte
          // LDS access (and, after the block loop, one 16-byte global store) instead of four 4-byte ones
          f32x4 px = {0.f, 0.f, 0.f, 0.f};
          auto head_part = [&](const int hh) __attribute__((always_inline)) {
            const Elem* th = tiles0 + (wave_of(hh, qs_) * 6 + 2 + 2 * par) * TS;
            if constexpr (DXL) {
              px = mma<F16>(wt.arole(0, hh, cc, lane), row_read<F16>(th, c, g), px);
              if (has_res) px = mma<F16>(wt.arole(2, hh, cc, lane), row_read<F16>(th + TS, c, g), px);
            } else {
              px = mma<F16>(row_read<F16>(th, c, g), wt.arole(0, hh, cc, lane), px);
              if (has_res) px = mma<F16>(row_read<F16>(th + TS, c, g), wt.arole(2, hh, cc, lane), px);
            }
          };
          if (NH == 4) {      // (the common head count, straight-line: all eight operand reads in flight before the first product)
            head_part(0); head_part(1); head_part(2); head_part(3);
          } else {
            for (int hh = 0; hh < NH; ++hh) head_part(hh);
          }
          const int kin = 16 * cc + c;
          if constexpr (DXL) {
            *reinterpret_cast<f32x4*>(dxs + (cc * d.FP + 16 * bi + c) * 16 + 4 * g) = px;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int f = 16 * bi + 4 * g + r;
              buf_store1(r_dx, (f < d.F && kin < d.K) ? 4 * x_off(d, b, f, kin) : kOOB, px[r]);      // [query 4g+r][kin c]
            }
          }
        }
      }
    };
    {
      int rotw = 0;
      for (int st = 0; st * WPH < d.nblk; ++st) {
        const int i = st * WPH + sub;
        if (WPH == 1 || i < d.nblk) {     // (wave-uniform; with two waves per head and an odd block count the last step is one wave's)
          phase_p(i);
          FIL_STAMP(1)
          phase_t();
          FIL_STAMP(2)
          phase_e(i, st & 1);
          // The next block's inputs (requested in P) are waited for HERE, where they have long arrived.  Left to the compiler the
          // wait lands at their first use in the next block's P, behind the loads that block issues for ITS successor
          // (SlabLane::load has two forms, and across that branch the wait count degrades to vmcnt(0)): every block then waited
          // for a full HBM round trip of loads it had just issued.
          settle4(n_dy);
          settle4(n_y);
          settle4(n_av);
          asm volatile("" : "+v"(n_rs));
          FIL_STAMP(3)
        }
        lds_barrier();   // every wave's dq / dres tile of this step is in LDS
        FIL_STAMP(4)
        phase_x(st, rotw);
        rotw = rotw + 1 == nw ? 0 : rotw + 1;
        FIL_STAMP(5)
      }
    }
    // the next sample's x rows on their way into L2 while this one finishes (its staging is a serial HBM round trip otherwise:
    // 12 % of the K = 64 kernel by the phase stamps, with the whole workgroup -- the only one on its CU -- waiting)
    if constexpr (XL) prefetch_x_lines<NC>(x, x_scrap, d, b + (int)gridDim.x, blockDim.x);
    // ---- dk of this head: accumulators -> the (now dead) k image as row fragments [key][a]
    // (two waves per head: each holds the sum over its own query blocks; the second wave's part goes through the image)
    if constexpr (WPH > 1) {
      // (the merge is split by key tile: wave `sub` hands its partner the tiles of the other parity, then adds the partner's part to
      // its own tiles and leaves the sums in the image -- half the chain of 13 round trips per wave; one wave writing everything and
      // the other reading, adding and writing everything back was 7 % of the K = 64 kernel)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < d.nblk && (j & 1) != sub) row_write<F16>(kimg, 16 * j + c, g, to_op<F16>(dk[j]));
      lds_barrier();
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < d.nblk && (j & 1) == sub) {
          dk[j] += from_op<F16>(row_read<F16>(kimg, 16 * j + c, g));
          row_write<F16>(kimg, 16 * j + c, g, to_op<F16>(dk[j]));
        }
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < d.nblk) row_write<F16>(kimg, 16 * j + c, g, to_op<F16>(dk[j]));
    }
    if constexpr (!DXL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the X phases' dx stores, before other waves read them back
    lds_barrier();
    FIL_STAMP_POST(2)      // dk merge
    for (int j = sub; j < d.nblk; j += WPH) {
      const Op dk_c = tr_read<F16>(kimg, 16 * j, lane);       // dk[key 4g+s][a c]
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) dWk[cc] = mma<F16>(xsrc.col(j, cc, lane), dk_c, dWk[cc]);
    }
    if constexpr (XL) {
      const int bn = b + (int)gridDim.x;
      stage_x_load<NC, true>(make_rsrc(x, (long)d.B * d.F * d.K * 4), d, bn, threadIdx.x, blockDim.x, x_early && bn < d.B, xpre);
    }
    FIL_STAMP_POST(3)      // dWk
    // dx += dk Wk^T (summed over the heads): jobs (block j, chunk cc), q = j NC + cc, dealt to the waves round robin
    {
      const int njobs = d.nblk * NC;
      if constexpr (DXL) {
        // (the dq part is in the LDS image as transposed tiles: a lane reads its four kin of one row, adds the dk part and stores the
        // finished 16 bytes straight to dx -- 1 KiB contiguous per wave instruction in the head-major layout; no write-back, no
        // barrier, no separate pass over the image, and the stores leave spread over the jobs instead of in one burst.
        // Two jobs per pass, interleaved: a job is a chain of LDS read -> H dependent products -> store, and a wave has 6-7 of them;
        // the second job of the last pass repeats the first and is not stored)
        const bool vecx = (d.xcw & 3) == 0 && (d.K & 3) == 0;
        // (with XFIX every job of a wave is chunk w % 4 here too; keeping its four Wk fragments in registers as well was measured:
        // 3 spilled registers, 535 -> 547 us)
        for (int q = w; q < njobs; q += 2 * nw) {
          const bool two = q + nw < njobs;
          const int qB = two ? q + nw : q;
          const int jA = q / NC, ccA = q - jA * NC, jB = qB / NC, ccB = qB - jB * NC;
          f32x4 pxA = *reinterpret_cast<const f32x4*>(dxs + (ccA * d.FP + 16 * jA + c) * 16 + 4 * g);
          f32x4 pxB = *reinterpret_cast<const f32x4*>(dxs + (ccB * d.FP + 16 * jB + c) * 16 + 4 * g);
          auto head_part = [&](const int hh) __attribute__((always_inline)) {
            pxA = mma<F16>(wt.arole(1, hh, ccA, lane), row_read<F16>(kimg0 + hh * KIS, 16 * jA + c, g), pxA);
            pxB = mma<F16>(wt.arole(1, hh, ccB, lane), row_read<F16>(kimg0 + hh * KIS, 16 * jB + c, g), pxB);
          };
          if (NH == 4) {
            head_part(0); head_part(1); head_part(2); head_part(3);
          } else {
            for (int hh = 0; hh < NH; ++hh) head_part(hh);
          }
          auto put = [&](const int j, const int cc, const f32x4& px, const bool on) __attribute__((always_inline)) {
            const int f = 16 * j + c, k0 = 16 * cc + 4 * g;
            if (vecx) {
              buf_store4(r_dx, (on && f < d.F && k0 < d.K) ? 4 * x_off(d, b, f, k0) : kOOB, px);
            } else {
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) buf_store1(r_dx, (on && f < d.F && k0 + s4 < d.K) ? 4 * x_off(d, b, f, k0 + s4) : kOOB, px[s4]);
            }
          };
          put(jA, ccA, pxA, true);
          put(jB, ccB, pxB, two);
        }
      } else {
        // (the dq part went to global memory: all read-backs of a batch of jobs are issued before the first is used -- they come
        // from L2 / the Infinity Cache; the dk accumulators' registers are free by now)
        constexpr int kBatch = 8;
        for (int q0 = w; q0 < njobs; q0 += kBatch * nw) {
          f32x4 old[kBatch];
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            const int q = q0 + u * nw, j = q / NC, cc = q - j * NC, kin = 16 * cc + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int f = 16 * j + 4 * g + r;
              old[u][r] = buf_load1(r_dx, (q < njobs && f < d.F && kin < d.K) ? 4 * x_off(d, b, f, kin) : kOOB);
            }
          }
#pragma unroll
          for (int u = 0; u < kBatch; ++u) {
            const int q = q0 + u * nw, j = q / NC, cc = q - j * NC, kin = 16 * cc + c;
            if (q < njobs) {
              f32x4 px = old[u];
              for (int hh = 0; hh < NH; ++hh)
                px = mma<F16>(row_read<F16>(kimg0 + hh * KIS, 16 * j + c, g), wt.arole(1, hh, cc, lane), px);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int f = 16 * j + 4 * g + r;
                buf_store1(r_dx, (f < d.F && kin < d.K) ? 4 * x_off(d, b, f, kin) : kOOB, px[r]);
              }
            }
          }
        }
      }
    }
    FIL_STAMP_POST(4)      // dx += dk Wk^T
  }
  FIL_STAMP(6)
#ifdef FIL_ATTN_STAMPS
  if (stamps != nullptr && lane == 0)
    for (int p8 = 0; p8 < 8; ++p8) stamps[((long)blockIdx.x * nw + w) * 8 + p8] = ph[p8];
#endif
  // ---- per-workgroup partials of the parameter gradients
  float* wp = wpart + ((long)blockIdx.x * WPH + sub) * 3 * d.K * d.H * d.A;   // one partial set per (workgroup, sub)
  const long wstride = (long)d.K * d.H * d.A;
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kin = 16 * cc + 4 * g + r;
      if (kin < d.K && c < d.A) {
        const long o = ((long)kin * d.H + h) * d.A + c;
        wp[o] = dWq[cc][r];
        wp[wstride + o] = dWk[cc][r];
        wp[2 * wstride + o] = dWr[cc][r];
      }
    }
  if (use_ln) {
    // lane (g,c) holds partial sums for a = 4g+s over its queries: sum the 16 lanes of the row
    float* gp = gb_part + ((long)blockIdx.x * nw + w) * 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float tg = row16_allsum(dgam[s]), tb = row16_allsum(dbet[s]);
      if (c == 0) {
        gp[4 * g + s] = tg;
        gp[16 + 4 * g + s] = tb;
      }
    }
  }
}

#if FIL_ATTN_PART == 0
// One launch sums every per-workgroup partial of the backward in a fixed order (1024 threads = 16 waves per workgroup):
//   workgroups [0, 3 nx):   out_j[i] = sum_p part[(p 3 + j) n + i], 64 outputs each (nx = ceil(n / 64)); wave v takes the partials
//                           v, v + 16, ... (four independent sums), the 16 wave sums are added in wave order
//   workgroups [3 nx, +32): dgamma / dbeta [a] = sum over the [blocks][2][16] partials, strided sums + a fixed tree
// (rounds 1-4: two launches of 256 threads; 16 us of a 0.8 ms layer step, half of it the second launch's latency)
__global__ __launch_bounds__(1024) void attn_reduce_kernel(const float* __restrict__ part, float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ o2, int n, int parts, const float* __restrict__ gb_part,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int gb_blocks, int A) {
  __shared__ float red[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nx = (n + 63) / 64;
  if ((int)blockIdx.x < 3 * nx) {
    const int j = blockIdx.x / nx, i = (blockIdx.x - j * nx) * 64 + lane;
    float* out = j == 0 ? o0 : (j == 1 ? o1 : o2);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (i < n) {
      const float* p0 = part + (long)j * n + i;
      const long ps = 3L * n;
      int p = wave;
      for (; p + 48 < parts; p += 64) {
        t0 += p0[(long)p * ps];
        t1 += p0[(long)(p + 16) * ps];
        t2 += p0[(long)(p + 32) * ps];
        t3 += p0[(long)(p + 48) * ps];
      }
      for (; p < parts; p += 16) t0 += p0[(long)p * ps];
    }
    red[wave * 64 + lane] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (wave == 0 && i < n && out != nullptr) {
      float t = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) t += red[v * 64 + lane];
      out[i] = t;
    }
  } else {
    if (dgamma == nullptr) return;
    const int q = blockIdx.x - 3 * nx, which = q >> 4, a = q & 15;
    float t = 0.f;
    for (int p = threadIdx.x; p < gb_blocks; p += 1024) t += gb_part[((long)p * 2 + which) * 16 + a];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
      if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
      __syncthreads();
    }
    if (threadIdx.x == 0 && a < A) (which == 0 ? dgamma : dbeta)[a] = red[0];
  }
}

#endif

// ------------------------------------------------------------------------------------------------- host
static int make_dims(const char* fn, int B, int F, int K, int H, int A, int x_chunk, AttnDims& d) {
  if (B < 0 || F < 1 || K < 1 || H < 1 || A < 1) return fail(FIL_ERR_ARG, "%s: bad shape B=%d F=%d K=%d H=%d A=%d", fn, B, F, K, H, A);
  if (K > 16 * kMaxNC) return fail(FIL_ERR_UNSUPPORTED, "%s: K=%d > %d", fn, K, 16 * kMaxNC);
  if (A > 16) return fail(FIL_ERR_UNSUPPORTED, "%s: attention_dim A=%d > 16", fn, A);
  if (H > kMaxHeads) return fail(FIL_ERR_UNSUPPORTED, "%s: H=%d > %d heads", fn, H, kMaxHeads);
  if (F > 512) return fail(FIL_ERR_UNSUPPORTED, "%s: F=%d > 512 fields", fn, F);
  if (x_chunk < 0 || (x_chunk > 0 && K % x_chunk != 0)) return fail(FIL_ERR_ARG, "%s: x_chunk=%d does not divide K=%d", fn, x_chunk, K);
  d.B = B; d.F = F; d.K = K; d.H = H; d.A = A;
  d.nblk = cdiv(F, 16);
  d.FP = 16 * d.nblk;
  d.NC = cdiv(K, 16);
  if ((long)B * F * K >= (1L << 29)) return fail(FIL_ERR_UNSUPPORTED, "%s: B*F*K = %ld >= 2^29 (32-bit buffer offsets)", fn, (long)B * F * K);
  d.xcw = x_chunk > 0 ? x_chunk : K;
  d.xcs = x_chunk > 0 ? B * F * x_chunk : 0;
  d.xrcp = x_chunk > 0 ? (65536 + x_chunk - 1) / x_chunk : 0;
  return FIL_OK;
}

static size_t fwd_lds(const AttnDims& d, bool f16) {
  return f16 ? ((size_t)d.NC * d.FP * 16 + (size_t)d.H * d.FP * 16) * sizeof(_Float16) : (size_t)d.H * d.FP * 20 * sizeof(float);
}
static size_t bwd_dx_lds(const AttnDims& d) { return (size_t)d.NC * d.FP * 16 * sizeof(float); }   // the sample's dx image
static size_t bwd_lds(const AttnDims& d, bool f16, int wph = 1, bool dx_img = false) {
  const size_t tiles = (size_t)d.H * wph * 6 + 3 * (size_t)d.H * d.NC;   // per-wave tiles + the weight table
  const size_t dxb = dx_img ? bwd_dx_lds(d) : 0;
  if (f16) {
    const size_t ximg = d.NC <= FIL_ATTN_XL_MAXNC ? (size_t)d.NC * d.FP * 16 : 0;
    return (ximg + (size_t)d.H * (d.FP + 16) * 16 + tiles * 256) * sizeof(_Float16) + dxb;
  }
  return ((size_t)d.H * (d.FP + 16) * 20 + (size_t)d.H * wph * 6 * 320) * sizeof(float) + dxb;   // no weight table in the f32 mode
}
// dynamic LDS a backward / forward launch may ask for: the CU's 160 KiB less the kernels' static arrays (gamma_s, x_scrap: 320 bytes),
// rounded to the allocation granule -- a shape within 320 bytes of the limit must take the smaller configuration, not fail at launch
constexpr size_t kLdsCap = 160 * 1024 - 512;
constexpr int kMaxBwdGrid = 1024;   // persistent workgroups (the workspace holds this many partial sums)

// persistent backward grid: `per_cu` resident workgroups on each of the 256 CUs, sized so that every workgroup takes the
// same number of samples (+-1)
static int bwd_grid(const AttnDims& d, int per_cu, int wph = 1) {
  const long cap = std::min<long>(kMaxBwdGrid / wph, 256L * std::max(1, per_cu));   // (grid * wph partial sets fit the workspace)
  const long rounds = ((long)d.B + cap - 1) / cap;
  return (int)std::max<long>(1, ((long)d.B + rounds - 1) / std::max<long>(rounds, 1));
}

static size_t attn_bwd_ws(const AttnDims& d, bool have_saved) {
  const long G = std::min<long>(kMaxBwdGrid, 2L * std::max(d.B, 1));   // partial sets: workgroups x waves per head
  size_t t = align_up((size_t)G * 3 * d.K * d.H * d.A * sizeof(float), 256);       // dW partials
  t += align_up((size_t)G * d.H * 32 * sizeof(float), 256);                         // dgamma/dbeta partials
  if (!have_saved)                                                                  // av, y and rstd recomputed
    t += 2 * align_up((size_t)d.H * d.B * d.F * d.A * sizeof(float), 256) + align_up((size_t)d.H * d.B * d.F * sizeof(float), 256);
  return t;
}

template <typename KernelT>
static int allow_lds_attn(KernelT kernel, size_t sh) {
  if (sh > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
      (void)hipGetLastError();
      return FIL_ERR_HIP;
    }
  }
  return FIL_OK;
}

// resident workgroups per CU of a kernel at this block size / dynamic LDS size (register-, LDS- and wave-limited);
// asked once per (kernel, launch shape) and remembered: the launch path stays free of runtime queries
template <typename KernelT>
static int resident_blocks(KernelT kernel, int threads, size_t sh) {
  struct Memo { const void* k; int threads; size_t sh; int n; };
  static Memo memo[64];
  static int used = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  const void* key = reinterpret_cast<const void*>(kernel);
  for (int i = 0; i < used; ++i)
    if (memo[i].k == key && memo[i].threads == threads && memo[i].sh == sh) return memo[i].n;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, sh) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    n = 1;
  }
  if (used < 64) memo[used++] = Memo{key, threads, sh, n};
  return n;
}

// CALL(NC, F16) for the runtime (NC, precision) pair; `f16` must be in scope
#ifdef FIL_ATTN_DEV
// experiment builds (tools/abl_build.py ... -DFIL_ATTN_DEV): only the instantiations BASELINE config 5 runs (f16, K = 16 and 64,
// F = 200) -- a tenth of the compile time; everything else falls through to "unsupported"
#define FIL_ATTN_NC(NCV, CALL)                    \
  switch (NCV) {                                  \
    case 1: { if (f16) { CALL(1, true); } } break; \
    case 4: { if (f16) { CALL(4, true); } } break; \
  }
#else
#define FIL_ATTN_NC(NCV, CALL)                                               \
  switch (NCV) {                                                             \
    case 1: { if (f16) { CALL(1, true); } else { CALL(1, false); } } break;  \
    case 2: { if (f16) { CALL(2, true); } else { CALL(2, false); } } break;  \
    case 3: { if (f16) { CALL(3, true); } else { CALL(3, false); } } break;  \
    case 4: { if (f16) { CALL(4, true); } else { CALL(4, false); } } break;  \
  }
#endif


// ---- the backward's instantiations, one translation unit per precision (attn_bwd_f16.hip / attn_bwd_f32.hip: they compile beside
// attn.hip instead of behind it -- the f16 set alone is 80 kernels)
struct AttnBwdLaunch {
  const float *x, *Wq, *Wk, *Wr, *gamma, *dy, *dres_in, *y_saved, *av_saved, *rstd_saved;
  float *dx, *wpart, *gb_part;
  AttnDims d;
  float scale, eps;
  int fuse_relu;
  long long* stamps;
  int wph, dx_lds;
  size_t sh;
  hipStream_t st;
};
int attn_launch_bwd_f16(const AttnBwdLaunch& a, int* G_out);   // FIL_OK, or FIL_ERR_HIP when the LDS reservation fails; *G_out = grid
int attn_launch_bwd_f32(const AttnBwdLaunch& a, int* G_out);

#if FIL_ATTN_PART == 1 || FIL_ATTN_PART == 2
#if FIL_ATTN_PART == 1
#define FIL_ATTN_P true
int attn_launch_bwd_f16(const AttnBwdLaunch& a, int* G_out) {
#else
#define FIL_ATTN_P false
int attn_launch_bwd_f32(const AttnBwdLaunch& a, int* G_out) {
#endif
  const float *x = a.x, *Wq = a.Wq, *Wk = a.Wk, *Wr = a.Wr, *gamma = a.gamma, *dy = a.dy, *dres_in = a.dres_in, *y_saved = a.y_saved,
              *av_saved = a.av_saved, *rstd_saved = a.rstd_saved;
  float *dx = a.dx, *wpart = a.wpart, *gb_part = a.gb_part;
  const AttnDims& d = a.d;
  const float scale = a.scale, eps = a.eps;
  const int fuse_relu = a.fuse_relu, wph = a.wph, dx_lds = a.dx_lds, H = a.d.H;
  long long* g_attn_stamps = a.stamps;
  const size_t sh = a.sh;
  hipStream_t st = a.st;
  const dim3 block(64 * H * wph);
  int lrc = FIL_OK, G = 1;
  (void)g_attn_stamps;
#if defined(FIL_ATTN_DEV) && FIL_ATTN_PART == 2
  lrc = FIL_ERR_HIP;      // (experiment builds carry the f16 instantiations only)
#else
// CALL(NC) for the runtime NC, precision fixed by the translation unit
#ifdef FIL_ATTN_DEV
#define FIL_ATTN_NC_P(NCV, CALL)              \
  switch (NCV) {                              \
    case 1: { CALL(1, FIL_ATTN_P); } break;   \
    case 4: { CALL(4, FIL_ATTN_P); } break;   \
  }
#else
#define FIL_ATTN_NC_P(NCV, CALL)              \
  switch (NCV) {                              \
    case 1: { CALL(1, FIL_ATTN_P); } break;   \
    case 2: { CALL(2, FIL_ATTN_P); } break;   \
    case 3: { CALL(3, FIL_ATTN_P); } break;   \
    case 4: { CALL(4, FIL_ATTN_P); } break;   \
  }
#endif
#define CALL_BWD_NBDA(N, P, NBV, WV, DX, AV)                                                                                        \
  lrc = allow_lds_attn(attn_bwd_kernel<N, P, NBV, WV, DX, AV>, sh);                                                                 \
  if (lrc == FIL_OK) {                                                                                                              \
    G = bwd_grid(d, resident_blocks(attn_bwd_kernel<N, P, NBV, WV, DX, AV>, 64 * H * WV, sh), WV);                                  \
    hipLaunchKernelGGL((attn_bwd_kernel<N, P, NBV, WV, DX, AV>), dim3(G), block, sh, st, x, Wq, Wk, Wr, gamma, dy, dres_in, y_saved, \
                       av_saved, rstd_saved, dx, wpart, gb_part, d, scale, eps, fuse_relu, g_attn_stamps);                         \
  }
// (the attention_dim == 16 form -- one 16-byte access per row piece, no column masks -- exists for the large-F instantiations, where
// the per-block prologue is a measurable part of the kernel; small shapes take the general form)
#define CALL_BWD_NBD(N, P, NBV, WV, DX) \
  if (NBV >= 13 && d.A == 16) { CALL_BWD_NBDA(N, P, NBV, WV, DX, (NBV >= 13)); } else { CALL_BWD_NBDA(N, P, NBV, WV, DX, false); }
#define CALL_BWD_NB(N, P, NBV, WV) \
  if (dx_lds) { CALL_BWD_NBD(N, P, NBV, WV, true); } else { CALL_BWD_NBD(N, P, NBV, WV, false); }
  // (the two-waves-per-head form exists for the large-F instantiations only: smaller shapes fit two workgroups per CU)
#ifdef FIL_ATTN_DEV
#define CALL_BWD(N, P) \
  if (d.nblk > 8 && d.nblk <= 13 && dx_lds) { if (wph == 2) { CALL_BWD_NBD(N, P, 13, 2, true); } else { CALL_BWD_NBD(N, P, 13, 1, true); } }
#else
#define CALL_BWD(N, P)                                                                     \
  if (d.nblk <= 4) { CALL_BWD_NB(N, P, 4, 1); }                                            \
  else if (d.nblk <= 8) { CALL_BWD_NB(N, P, 8, 1); }                                       \
  else if (d.nblk <= 13) { if (wph == 2) { CALL_BWD_NB(N, P, 13, 2); } else { CALL_BWD_NB(N, P, 13, 1); } } \
  else { if (wph == 2) { CALL_BWD_NB(N, P, 32, 2); } else { CALL_BWD_NB(N, P, 32, 1); } }
#endif
    FIL_ATTN_NC_P(d.NC, CALL_BWD)
#undef CALL_BWD
#undef CALL_BWD_NB
#undef CALL_BWD_NBD
#undef CALL_BWD_NBDA
#undef FIL_ATTN_NC_P
#endif
  *G_out = G;
  return lrc;
}
#undef FIL_ATTN_P
#endif   // FIL_ATTN_PART 1, 2

}  // namespace fil

#if FIL_ATTN_PART == 0
namespace fil {
// diagnostic builds (-DFIL_ATTN_STAMPS): device buffer the backward writes its per-phase clocks to (fil_attn_debug_stamps)
static long long* g_attn_stamps = nullptr;

static int launch_fwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma, const float* beta,
                      float* y, float* res_out, float* av_out, float* rstd_out, const AttnDims& d, float scale, float eps,
                      int fuse_relu, bool f16, hipStream_t st) {
  // f16 mode, up to 13 key tiles: the k fragments live in registers (no k image in LDS; FIL_ATTN_KREG=0 keeps the image)
  static const int kreg_knob = [] {
    const char* e = getenv("FIL_ATTN_KREG");
    return e != nullptr ? atoi(e) : 1;
  }();
  const bool kreg = f16 && d.nblk <= 13 && kreg_knob != 0;
  const size_t sh = kreg ? (size_t)d.NC * d.FP * 16 * sizeof(_Float16) : fwd_lds(d, f16);
  if (sh > kLdsCap) return fail(FIL_ERR_UNSUPPORTED, "fil_attn_fwd: F=%d K=%d H=%d needs %zu bytes of LDS (> 160 KiB)", d.F, d.K, d.H, sh);
  const dim3 grid(d.B), block(64 * d.H);
  int rc = FIL_OK;
#define CALL_FWD_K(...)                                                                                                       \
  rc = allow_lds_attn(attn_fwd_kernel<__VA_ARGS__>, sh);                                                                      \
  if (rc == FIL_OK)                                                                                                           \
    hipLaunchKernelGGL((attn_fwd_kernel<__VA_ARGS__>), grid, block, sh, st, x, Wq, Wk, Wr, gamma, beta, y, res_out, av_out,  \
                       rstd_out, d, scale, eps, fuse_relu)
#define CALL_FWD_A(N, P, AV) \
  if (kreg) { CALL_FWD_K(N, P, AV, (P ? 13 : 0)); } else { CALL_FWD_K(N, P, AV, 0); }
#define CALL_FWD(N, P) \
  if (d.A == 16) { CALL_FWD_A(N, P, true); } else { CALL_FWD_A(N, P, false); }
  FIL_ATTN_NC(d.NC, CALL_FWD)
#undef CALL_FWD
#undef CALL_FWD_A
#undef CALL_FWD_K
  if (rc != FIL_OK) return fail(rc, "fil_attn_fwd: cannot reserve %zu bytes of LDS", sh);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

}  // namespace fil

using namespace fil;

#ifdef FIL_ATTN_STAMPS
extern "C" void fil_attn_debug_stamps(long long* buf) { g_attn_stamps = buf; }
#endif

extern "C" size_t fil_attn_fwd_workspace_bytes(int, int, int, int, int) { return 0; }

extern "C" size_t fil_attn_bwd_workspace_bytes(int B, int F, int K, int H, int A, int have_saved) {
  AttnDims d;
  if (make_dims("fil_attn_bwd_workspace_bytes", B, F, K, H, A, 0, d) != FIL_OK || B == 0) return 0;
  return attn_bwd_ws(d, have_saved != 0);
}

extern "C" int fil_attn_fwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                            const float* beta, float* y, float* res_out, float* av_out, float* rstd_out, int B, int F, int K,
                            int H, int A, float scale, float eps, int fuse_relu, int precision, int x_chunk, void* workspace,
                            size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  AttnDims d;
  int rc = make_dims("fil_attn_fwd", B, F, K, H, A, x_chunk, d);
  if (rc != FIL_OK) return rc;
  if (precision != FIL_PREC_F32 && precision != FIL_PREC_F16_MFMA) return fail(FIL_ERR_ARG, "fil_attn_fwd: precision=%d", precision);
  const bool f16 = precision == FIL_PREC_F16_MFMA;
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(x && Wq && Wk && y);
  FIL_CHECK_ARG((gamma == nullptr) == (beta == nullptr));
  FIL_CHECK_ARG((av_out == nullptr) == (rstd_out == nullptr));   // kept together or not at all
  hipStream_t st = (hipStream_t)stream;
  // algorithmic flops: projections 2*F*K*A*(2 or 3) + scores and weighted sum 2*2*F*F*A, per (b,h)
  ProfScope ps("attn_fwd", st, (double)B * H * (2.0 * F * K * A * (Wr ? 3 : 2) + 4.0 * F * (double)F * A));
  return launch_fwd(x, Wq, Wk, Wr, gamma, beta, y, res_out, av_out, rstd_out, d, scale, eps, fuse_relu, f16, st);
}

extern "C" int fil_attn_bwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                            const float* beta, const float* dy, const float* dres_in, const float* y_saved,
                            const float* av_saved, const float* rstd_saved, float* dx, float* dWq, float* dWk, float* dWr,
                            float* dgamma, float* dbeta, int B, int F, int K, int H, int A, float scale, float eps, int fuse_relu, int precision,
                            int x_chunk, void* workspace, size_t workspace_bytes, void* stream) {
  AttnDims d;
  int rc = make_dims("fil_attn_bwd", B, F, K, H, A, x_chunk, d);
  if (rc != FIL_OK) return rc;
  if (precision != FIL_PREC_F32 && precision != FIL_PREC_F16_MFMA) return fail(FIL_ERR_ARG, "fil_attn_bwd: precision=%d", precision);
  const bool f16 = precision == FIL_PREC_F16_MFMA;
  FIL_CHECK_ARG(Wq && Wk && dWq && dWk);
  FIL_CHECK_ARG((gamma == nullptr) == (beta == nullptr));
  FIL_CHECK_ARG(gamma == nullptr || (dgamma && dbeta));
  FIL_CHECK_ARG(Wr == nullptr || dWr != nullptr);
  FIL_CHECK_ARG((av_saved == nullptr) == (rstd_saved == nullptr));
  hipStream_t st = (hipStream_t)stream;
  const size_t wsz = (size_t)K * H * A * sizeof(float);
  if (B == 0) {
    (void)hipMemsetAsync(dWq, 0, wsz, st);
    (void)hipMemsetAsync(dWk, 0, wsz, st);
    if (dWr) (void)hipMemsetAsync(dWr, 0, wsz, st);
    if (dgamma) (void)hipMemsetAsync(dgamma, 0, A * sizeof(float), st);
    if (dbeta) (void)hipMemsetAsync(dbeta, 0, A * sizeof(float), st);
    return FIL_OK;
  }
  FIL_CHECK_ARG(x && dy && dx);
  // the LayerNorm backward needs av, the fused ReLU mask needs y: whatever the caller did not keep is recomputed
  const bool need_av = gamma != nullptr && av_saved == nullptr, need_y = fuse_relu && y_saved == nullptr;
  const size_t need_ws = attn_bwd_ws(d, !(need_av || need_y));
  if (workspace == nullptr || workspace_bytes < need_ws)
    return fail(FIL_ERR_WORKSPACE, "fil_attn_bwd: workspace %zu < %zu bytes", workspace_bytes, need_ws);
  // unfused mode: the residual branch's gradient arrives separately (dres_in)
  const bool has_res = Wr != nullptr;
  if (!fuse_relu && has_res && dres_in == nullptr) return fail(FIL_ERR_ARG, "fil_attn_bwd: dres is required when fuse_relu == 0 and Wr != NULL");
  size_t sh = bwd_lds(d, f16);
  if (sh > kLdsCap)
    return fail(FIL_ERR_UNSUPPORTED, "fil_attn_bwd: F=%d K=%d H=%d needs %zu bytes of LDS (> 160 KiB)%s", F, K, H, sh,
                f16 ? "" : "; the f16-MFMA precision needs less than half of that");
  // two waves per head when the footprint lets only ONE one-wave-per-head workgroup onto a CU (and the doubled workgroup fits)
  static const int wph_knob = [] {   // 0 = automatic, 1 / 2 = forced (tests, sweeps); read once
    const char* e = getenv("FIL_ATTN_WPH");
    return e != nullptr ? atoi(e) : 0;
  }();
  int wph = (2 * sh > kLdsCap && d.nblk > 8) ? 2 : 1;
  // ... and also where two one-wave-per-head workgroups do fit but only WITHOUT the LDS dx image, while one two-waves-per-head
  // workgroup fits with it: the image is worth more than the second workgroup (K = 64, F = 160, f16: 0.435 ms against 0.462 --
  // tools/attn_occ_probe.py; the dx part through global memory is a read-modify-write of 2 x |dx| per launch)
  if (wph == 1 && d.nblk > 8 && H <= 4) {
    const size_t sh1d = bwd_lds(d, f16, 1, true), sh2d = bwd_lds(d, f16, 2, true);
    const bool dx_with_1 = sh1d <= kLdsCap && kLdsCap / sh1d >= std::min<size_t>(kLdsCap / sh, 2);
    if (!dx_with_1 && sh2d <= kLdsCap) wph = 2;
  }
  if (wph_knob == 1 || wph_knob == 2) wph = d.nblk > 8 ? wph_knob : 1;
  if (wph == 2 && (H > 4 || bwd_lds(d, f16, 2) > kLdsCap || d.nblk < 2)) wph = 1;
  sh = bwd_lds(d, f16, wph);
  // dx image in LDS when it fits without costing a resident workgroup (FIL_ATTN_DX_LDS=0 keeps the global second visit)
  static const int dxl_knob = [] {
    const char* e = getenv("FIL_ATTN_DX_LDS");
    return e != nullptr ? atoi(e) : 1;
  }();
  int dx_lds = 0;
  if (dxl_knob != 0) {
    const size_t sh2 = bwd_lds(d, f16, wph, true);
    if (sh2 <= kLdsCap && kLdsCap / sh2 >= std::min<size_t>(kLdsCap / sh, 2)) {   // (at most two workgroups per CU run anyway: registers)
      dx_lds = 1;
      sh = sh2;
    }
  }
  const long Gws = std::min<long>(kMaxBwdGrid, 2L * B);
  Carver ws(workspace);
  float* wpart = ws.take<float>((size_t)Gws * 3 * K * H * A);
  float* gb_part = ws.take<float>((size_t)Gws * H * 32);
  int G = 1;
  const size_t nact = (size_t)H * B * F * A;
  if (need_av || need_y) {
    float* av_ws = ws.take<float>(nact);
    float* y_ws = ws.take<float>(nact);
    ProfScope ps("attn_bwd_recompute", st, (double)B * H * (2.0 * F * K * A * 3 + 4.0 * F * (double)F * A));
    float* rstd_ws = ws.take<float>((size_t)H * B * F);
    rc = launch_fwd(x, Wq, Wk, Wr, gamma, beta, y_ws, nullptr, need_av ? av_ws : nullptr, need_av ? rstd_ws : nullptr, d, scale, eps,
                    fuse_relu, f16, st);
    if (rc != FIL_OK) return rc;
    if (need_av) {
      av_saved = av_ws;
      rstd_saved = rstd_ws;
    }
    if (need_y) y_saved = y_ws;
  }
  {
    // algorithmic flops of the score pass: S, dS, dq, 2 x dk = 5 products of 2*F*F*A, plus projections and their gradients
    ProfScope ps("attn_bwd", st, (double)B * H * (10.0 * F * (double)F * A + 2.0 * F * K * A * (has_res ? 9 : 7)));
    int lrc = FIL_OK;
    const AttnBwdLaunch la{x, Wq, Wk, Wr, gamma, dy, dres_in, y_saved, av_saved, rstd_saved, dx, wpart, gb_part, d, scale, eps, fuse_relu, g_attn_stamps,
                           wph, dx_lds, sh, st};
    lrc = f16 ? attn_launch_bwd_f16(la, &G) : attn_launch_bwd_f32(la, &G);
    if (lrc != FIL_OK) return fail(lrc, "fil_attn_bwd: cannot reserve %zu bytes of LDS", sh);
    FIL_CHECK_LAUNCH();
  }
  const int n = K * H * A;
  hipLaunchKernelGGL(attn_reduce_kernel, dim3(3 * cdiv(n, 64) + (gamma != nullptr ? 32 : 0)), dim3(1024), 0, st, wpart, dWq, dWk, dWr, n,
                     G * wph, gb_part, gamma != nullptr ? dgamma : nullptr, dbeta, G * H * wph, A);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
#endif   // FIL_ATTN_PART == 0
