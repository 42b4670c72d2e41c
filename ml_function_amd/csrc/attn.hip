// A4  AutoInt interacting layer (multi-head field attention) for gfx950, fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Replaces MultHeadAttentionLayer.call + ProductAttentionLayer.call (behavior_layer.py:292-311,356-377) and the
// Add + ReLU of the DnnLayer wrapper (core_layer.py:204-216).  Reference quirks kept: the "softmax" is a sigmoid,
// V is projected with key_w (so K == V), LayerNorm eps is Keras' 1e-3, output is head-major [H,B,F,A].
// The [H,B,F,F] score tensor (2.6 GB at F=200, B=4096) is never materialised: a workgroup owns one (sample, head),
// projects x -> q, k in-kernel (K is tiny), keeps k in LDS and streams 16x16 score tiles through registers:
//     S' tile = sigmoid(scale * k_tile q_blk^T)   (accumulator layout: key f' on (lane>>4, reg), query f on lane&15)
//     av_blk += S'^T k_tile                       (the S' accumulator IS the next MFMA's A operand, no LDS round trip;
//                                                  the reduction order over f' is permuted to match: step s of lane group g
//                                                  takes f' = 16t + 4g + s)
// Backward recomputes the scores twice, once per orientation, so that both dq (reduce over keys) and dk (reduce
// over queries) take their A operand straight from accumulators:
//   attn_bwd_pre  recompute av, ReLU mask, LayerNorm backward -> dav, dres (global), dgamma/dbeta partials
//   attn_bwd_dq   S', dS' = k dav^T (key-major)   -> dq  = (dS' S'(1-S') scale)^T k
//   attn_bwd_dk   S,  dS  = dav k^T (query-major) -> dk  = (dS S(1-S) scale)^T q + S^T dav    (K and V share key_w)
//   attn_bwd_proj dx = dq Wq^T + dk Wk^T + dres Wr^T (summed over heads), dW* = x^T d*  (block partials + fixed-order reduce)
//
// MFMA 16x16x4 f32 maps: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r = D[row=4*(l>>4)+r][col=l&15].
#include "common.h"
#include <cstdlib>

namespace fil {

constexpr int kAttnThreads = 256;
constexpr int kRS = 20;       // LDS row stride of [rows][16] tiles: 16-byte aligned rows, conflict-free b128 row reads
constexpr int kMaxNC = 4;     // K <= 64 (NC = ceil(K/16) chunks of 16 along the projection's reduction)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Four reduction steps into one accumulator: lane group g = lane>>4 contributes the products a_s * b_s, s = 0..3.
// F16 = false: four exact-fp32 16x16x4 MFMAs (the 1e-5 parity mode).  F16 = true (BASELINE config 5, "fp16 MFMA"):
// operands rounded to fp16 (v_cvt_pk_f16_f32, round-to-nearest-even), ONE 16x16x16 f16 MFMA with fp32 accumulation --
// lane group g supplies k = 4g..4g+3 of both operands, which is the same (s, g) pairing as the four fp32 steps.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ f32x4 mma4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3, f32x4 c) {
  if constexpr (F16) {
    const f32x4 av = {a0, a1, a2, a3}, bv = {b0, b1, b2, b3};
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_convertvector(av, f16x4), __builtin_convertvector(bv, f16x4), c, 0, 0, 0);
  } else {
    c = mfma16(a0, b0, c);
    c = mfma16(a1, b1, c);
    c = mfma16(a2, b2, c);
    c = mfma16(a3, b3, c);
    return c;
  }
}

__device__ __forceinline__ float sigmoidf_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// sum over the 16 lanes that share lane>>4 (one row of a D tile)
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

struct AttnDims {
  int B, F, K, H, A;
  int nblk;   // ceil(F/16)
  int FP;     // 16*nblk
  int NC;     // ceil(K/16)
  int XSS;    // xs row stride = 16*NC + 4
};

// xs[f][k] = x[b,f,k], zero padded to [FP][16*NC]
__device__ __forceinline__ void stage_x(const float* __restrict__ xb, float* xs, const AttnDims& d) {
  const int Kpad = 16 * d.NC;
  for (int idx = threadIdx.x; idx < d.FP * Kpad; idx += kAttnThreads) {
    const int f = idx / Kpad, k = idx - f * Kpad;
    xs[f * d.XSS + k] = (f < d.F && k < d.K) ? xb[f * d.K + k] : 0.f;
  }
}

// w[c][s] = W[k = 16c + 4g + s][h][a = lane&15]  (zero for k >= K or a >= A)
template <int NC>
__device__ __forceinline__ void load_w(const float* __restrict__ W, int h, const AttnDims& d, int lane, float (&w)[NC][4]) {
  const int a = lane & 15, g = lane >> 4;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 16 * c + 4 * g + s;
      w[c][s] = (W != nullptr && k < d.K && a < d.A) ? W[((long)k * d.H + h) * d.A + a] : 0.f;
    }
}

// xr[c][s] = xs[16*blk + (lane&15)][16c + 4g + s]
template <int NC>
__device__ __forceinline__ void load_xfrag(const float* xs, int blk, const AttnDims& d, int lane, float (&xr)[NC][4]) {
  const float* p = xs + (16 * blk + (lane & 15)) * d.XSS + 4 * (lane >> 4);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float4 t = *reinterpret_cast<const float4*>(p + 16 * c);
    xr[c][0] = t.x; xr[c][1] = t.y; xr[c][2] = t.z; xr[c][3] = t.w;
  }
}

// rows form: D[row <-> f = 4g+r][col <-> a] = x_blk W
template <int NC, bool F16>
__device__ __forceinline__ f32x4 proj_rows(const float (&xr)[NC][4], const float (&w)[NC][4]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) acc = mma4<F16>(xr[c][0], xr[c][1], xr[c][2], xr[c][3], w[c][0], w[c][1], w[c][2], w[c][3], acc);
  return acc;
}

// transposed form: D[row <-> a = 4g+r][col <-> f] = (x_blk W)^T
template <int NC, bool F16>
__device__ __forceinline__ f32x4 proj_T(const float (&xr)[NC][4], const float (&w)[NC][4]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) acc = mma4<F16>(w[c][0], w[c][1], w[c][2], w[c][3], xr[c][0], xr[c][1], xr[c][2], xr[c][3], acc);
  return acc;
}

// store a rows-form tile into an LDS [FP][kRS] array
__device__ __forceinline__ void store_rows(float* arr, int blk, int lane, const f32x4& v) {
  float* p = arr + (16 * blk + 4 * (lane >> 4)) * kRS + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r * kRS] = v[r];
}

__device__ __forceinline__ float4 lds_row4(const float* arr, int row, int g) {
  return *reinterpret_cast<const float4*>(arr + row * kRS + 4 * g);
}

// all waves: arr[f][a] = (x W)[f][a] for every 16-row block (blocks dealt round-robin to the waves)
template <int NC, bool F16>
__device__ __forceinline__ void project_all(const float* xs, float* arr, const float (&w)[NC][4], const AttnDims& d, int wave,
                                            int lane) {
  for (int blk = wave; blk < d.nblk; blk += 4) {
    float xr[NC][4];
    load_xfrag<NC>(xs, blk, d, lane, xr);
    store_rows(arr, blk, lane, proj_rows<NC, F16>(xr, w));
  }
}

// av_blk = sum_t sigmoid(k_t q_blk^T)^T k_t  -- shared by forward and bwd_pre.  qT is pre-scaled.
template <bool F16>
__device__ __forceinline__ f32x4 attend_block(const float* kks, const f32x4& qT, int nblk, int lane) {
  const int a = lane & 15, g = lane >> 4;
  f32x4 av = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < nblk; ++t) {
    const float4 kA = lds_row4(kks, 16 * t + a, g);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f};
    sc = mma4<F16>(kA.x, kA.y, kA.z, kA.w, qT[0], qT[1], qT[2], qT[3], sc);
    const float* kb = kks + (16 * t + 4 * g) * kRS + a;
    av = mma4<F16>(sigmoidf_fast(sc[0]), sigmoidf_fast(sc[1]), sigmoidf_fast(sc[2]), sigmoidf_fast(sc[3]), kb[0], kb[kRS], kb[2 * kRS],
                   kb[3 * kRS], av);
  }
  return av;
}

struct LnOut {
  float xhat[4], rstd[4], ln[4];
};

// LayerNorm over a (the 16 lanes of a row), biased variance, only a < A counts
__device__ __forceinline__ void layer_norm_rows(const f32x4& av, bool avalid, float inv_a, float eps, float gam, float bet,
                                                bool use_ln, LnOut& o) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (use_ln) {
      const float v = avalid ? av[r] : 0.f;
      const float mu = row16_sum(v) * inv_a;
      const float dv = avalid ? v - mu : 0.f;
      const float var = row16_sum(dv * dv) * inv_a;
      const float rstd = 1.0f / sqrtf(var + eps);
      o.rstd[r] = rstd;
      o.xhat[r] = dv * rstd;
      o.ln[r] = o.xhat[r] * gam + bet;
    } else {
      o.rstd[r] = 1.f;
      o.xhat[r] = 0.f;
      o.ln[r] = av[r];
    }
  }
}

// ================================================================================================= forward
// y[h,b,f,a] = fuse_relu ? relu(res + ln) : ln ; res_out (optional, !fuse_relu) = x Wr
template <int NC, bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                                const float* __restrict__ Wk, const float* __restrict__ Wr,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ y, float* __restrict__ res_out,
                                                                float* __restrict__ av_out, AttnDims d, float scale, float eps,
                                                                int fuse_relu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                    // [FP][XSS]
  float* kks = smem + d.FP * d.XSS;    // [FP][kRS]
  const int b = blockIdx.x / d.H, h = blockIdx.x - b * d.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane & 15, g = lane >> 4;
  stage_x(x + (long)b * d.F * d.K, xs, d);
  float wq[NC][4], wk[NC][4], wr[NC][4];
  load_w<NC>(Wq, h, d, lane, wq);
  load_w<NC>(Wk, h, d, lane, wk);
  load_w<NC>(Wr, h, d, lane, wr);
  const bool use_ln = gamma != nullptr;
  const bool avalid = a < d.A;
  const float gam = (use_ln && avalid) ? gamma[a] : 0.f, bet = (use_ln && avalid) ? beta[a] : 0.f;
  const float inv_a = 1.0f / (float)d.A;
  __syncthreads();
  project_all<NC, F16>(xs, kks, wk, d, wave, lane);
  __syncthreads();

  for (int blk = wave; blk < d.nblk; blk += 4) {
    float xr[NC][4];
    load_xfrag<NC>(xs, blk, d, lane, xr);
    f32x4 qT = proj_T<NC, F16>(xr, wq);
#pragma unroll
    for (int r = 0; r < 4; ++r) qT[r] *= scale;
    const f32x4 av = attend_block<F16>(kks, qT, d.nblk, lane);
    LnOut ln;
    layer_norm_rows(av, avalid, inv_a, eps, gam, bet, use_ln, ln);
    f32x4 res = {0.f, 0.f, 0.f, 0.f};
    if (Wr != nullptr) res = proj_rows<NC, F16>(xr, wr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * blk + 4 * g + r;
      if (f < d.F && avalid) {
        const long o = (((long)h * d.B + b) * d.F + f) * d.A + a;
        if (av_out != nullptr) av_out[o] = av[r];   // saved for the backward (it then skips the score recomputation)
        if (fuse_relu) {
          y[o] = fmaxf(res[r] + ln.ln[r], 0.f);
        } else {
          y[o] = ln.ln[r];
          if (res_out != nullptr) res_out[o] = res[r];
        }
      }
    }
  }
}

// ================================================================================================= backward: pre
// Recomputes av; fused mode: dz = dy * (res + ln > 0), dres = dz; unfused: dz = dy (grad of ln), dres given separately.
// Writes dav (LayerNorm backward of dz) and, in fused mode, dres; accumulates dgamma/dbeta partials per workgroup.
template <int NC, bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_pre_kernel(
    const float* __restrict__ x, const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wr,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ dy, float* __restrict__ dav,
    float* __restrict__ dres, float* __restrict__ gb_part /* [blocks][2][16] */, AttnDims d, float scale, float eps,
    int fuse_relu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* kks = smem + d.FP * d.XSS;
  float* red = kks + d.FP * kRS;       // [4 waves][2][16]
  const int b = blockIdx.x / d.H, h = blockIdx.x - b * d.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane & 15, g = lane >> 4;
  stage_x(x + (long)b * d.F * d.K, xs, d);
  float wq[NC][4], wk[NC][4], wr[NC][4];
  load_w<NC>(Wq, h, d, lane, wq);
  load_w<NC>(Wk, h, d, lane, wk);
  load_w<NC>(Wr, h, d, lane, wr);
  const bool use_ln = gamma != nullptr;
  const bool avalid = a < d.A;
  const float gam = (use_ln && avalid) ? gamma[a] : 0.f, bet = (use_ln && avalid) ? beta[a] : 0.f;
  const float inv_a = 1.0f / (float)d.A;
  __syncthreads();
  project_all<NC, F16>(xs, kks, wk, d, wave, lane);
  __syncthreads();

  float dg = 0.f, db = 0.f;
  for (int blk = wave; blk < d.nblk; blk += 4) {
    float xr[NC][4];
    load_xfrag<NC>(xs, blk, d, lane, xr);
    f32x4 qT = proj_T<NC, F16>(xr, wq);
#pragma unroll
    for (int r = 0; r < 4; ++r) qT[r] *= scale;
    const f32x4 av = attend_block<F16>(kks, qT, d.nblk, lane);
    LnOut ln;
    layer_norm_rows(av, avalid, inv_a, eps, gam, bet, use_ln, ln);
    f32x4 res = {0.f, 0.f, 0.f, 0.f};
    if (fuse_relu && Wr != nullptr) res = proj_rows<NC, F16>(xr, wr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * blk + 4 * g + r;
      const bool valid = f < d.F && avalid;
      const long o = (((long)h * d.B + b) * d.F + f) * d.A + a;
      float dz = valid ? dy[o] : 0.f;
      if (fuse_relu) {
        if (!(res[r] + ln.ln[r] > 0.f)) dz = 0.f;
        if (valid && dres != nullptr) dres[o] = dz;
      }
      float da = dz;
      if (use_ln) {
        dg = fmaf(dz, ln.xhat[r], dg);
        db += dz;
        const float dxh = dz * gam;
        const float m1 = row16_sum(dxh) * inv_a;
        const float m2 = row16_sum(dxh * ln.xhat[r]) * inv_a;
        da = ln.rstd[r] * (dxh - m1 - ln.xhat[r] * m2);
      }
      if (valid) dav[o] = da;
    }
  }
  // dgamma/dbeta: lanes sharing a (4 groups) -> wave -> workgroup, fixed order
  dg += __shfl_xor(dg, 16); dg += __shfl_xor(dg, 32);
  db += __shfl_xor(db, 16); db += __shfl_xor(db, 32);
  if (lane < 16) {
    red[(wave * 2 + 0) * 16 + lane] = dg;
    red[(wave * 2 + 1) * 16 + lane] = db;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4, aa = threadIdx.x & 15;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * 16 + aa];
    gb_part[((long)blockIdx.x * 2 + which) * 16 + aa] = t;
  }
}

// ================================================================================================= backward: pre (saved av)
// Same outputs as attn_bwd_pre_kernel, from the forward's saved av (and, in fused mode, its output y: the ReLU mask
// is y > 0) instead of recomputing the scores: purely element-wise + 16-lane row reductions, memory-bound.
// 16 lanes per row [.., a < 16]; a workgroup walks rows with a grid stride; gb_part[block][2][16].
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_pre_saved_kernel(const float* __restrict__ av_s, const float* __restrict__ y_s,
                                                                         const float* __restrict__ gamma, const float* __restrict__ dy,
                                                                         float* __restrict__ dav, float* __restrict__ dres,
                                                                         float* __restrict__ gb_part, long rows, int A, float eps,
                                                                         int fuse_relu) {
  __shared__ float red[4][2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane & 15, g = lane >> 4;
  const bool use_ln = gamma != nullptr;
  const bool avalid = a < A;
  const float gam = (use_ln && avalid) ? gamma[a] : 0.f;
  const float inv_a = 1.0f / (float)A;
  float dg = 0.f, db = 0.f;
  const long rpb = 16;  // rows per workgroup pass: 4 waves x 4 lane groups
  for (long row0 = (long)blockIdx.x * rpb; row0 < rows; row0 += (long)gridDim.x * rpb) {
    const long row = row0 + wave * 4 + g;
    const bool valid = row < rows && avalid;
    const long o = row * A + a;
    float dz = valid ? dy[o] : 0.f;
    if (fuse_relu) {
      if (!(valid && y_s[o] > 0.f)) dz = 0.f;
      if (valid && dres != nullptr) dres[o] = dz;
    }
    float da = dz;
    if (use_ln) {
      const float v = valid ? av_s[o] : 0.f;
      const float mu = row16_sum(v) * inv_a;
      const float dv = valid ? v - mu : 0.f;
      const float var = row16_sum(dv * dv) * inv_a;
      const float rstd = 1.0f / sqrtf(var + eps);
      const float xhat = dv * rstd;
      dg = fmaf(dz, xhat, dg);
      db += dz;
      const float dxh = dz * gam;
      const float m1 = row16_sum(dxh) * inv_a;
      const float m2 = row16_sum(dxh * xhat) * inv_a;
      da = rstd * (dxh - m1 - xhat * m2);
    }
    if (valid) dav[o] = da;
  }
  dg += __shfl_xor(dg, 16); dg += __shfl_xor(dg, 32);
  db += __shfl_xor(db, 16); db += __shfl_xor(db, 32);
  if (lane < 16) {
    red[wave][0][lane] = dg;
    red[wave][1][lane] = db;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int which = threadIdx.x >> 4, aa = threadIdx.x & 15;
    gb_part[((long)blockIdx.x * 2 + which) * 16 + aa] = ((red[0][which][aa] + red[1][which][aa]) + red[2][which][aa]) + red[3][which][aa];
  }
}

// ================================================================================================= backward: dq
// key-major orientation: S'[f'][f], dS'[f'][f] = k[f'] . dav[f];  dq_blk = sum_t (dS' S'(1-S') scale)^T k_t
template <int NC, bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                                   const float* __restrict__ Wk, const float* __restrict__ dav,
                                                                   float* __restrict__ dq, AttnDims d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* kks = smem + d.FP * d.XSS;
  const int b = blockIdx.x / d.H, h = blockIdx.x - b * d.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane & 15, g = lane >> 4;
  stage_x(x + (long)b * d.F * d.K, xs, d);
  float wq[NC][4], wk[NC][4];
  load_w<NC>(Wq, h, d, lane, wq);
  load_w<NC>(Wk, h, d, lane, wk);
  __syncthreads();
  project_all<NC, F16>(xs, kks, wk, d, wave, lane);
  __syncthreads();
  const float* davh = dav + ((long)h * d.B + b) * d.F * d.A;
  float* dqh = dq + ((long)h * d.B + b) * d.F * d.A;

  for (int blk = wave; blk < d.nblk; blk += 4) {
    float xr[NC][4];
    load_xfrag<NC>(xs, blk, d, lane, xr);
    f32x4 qT = proj_T<NC, F16>(xr, wq);
#pragma unroll
    for (int r = 0; r < 4; ++r) qT[r] *= scale;
    // B operand of dS': lane (j = f = lane&15, k = g) needs dav[f][4g+s]
    float dv[4];
    {
      const int f = 16 * blk + a;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int aa = 4 * g + s;
        dv[s] = (f < d.F && aa < d.A) ? davh[(long)f * d.A + aa] : 0.f;
      }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < d.nblk; ++t) {
      const float4 kA = lds_row4(kks, 16 * t + a, g);
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, ds = {0.f, 0.f, 0.f, 0.f};
      sc = mma4<F16>(kA.x, kA.y, kA.z, kA.w, qT[0], qT[1], qT[2], qT[3], sc);
      ds = mma4<F16>(kA.x, kA.y, kA.z, kA.w, dv[0], dv[1], dv[2], dv[3], ds);
      const float* kb = kks + (16 * t + 4 * g) * kRS + a;
      float dpre[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float sg = sigmoidf_fast(sc[s]);
        dpre[s] = ds[s] * sg * (1.f - sg) * scale;
      }
      acc = mma4<F16>(dpre[0], dpre[1], dpre[2], dpre[3], kb[0], kb[kRS], kb[2 * kRS], kb[3 * kRS], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * blk + 4 * g + r;
      if (f < d.F && a < d.A) dqh[(long)f * d.A + a] = acc[r];
    }
  }
}

// ================================================================================================= backward: dk
// query-major orientation: S[f][f'], dS[f][f'] = dav[f] . k[f'];  waves split the key tiles.
//   dk_t = sum_blk (dS S(1-S) scale)^T q_blk + S^T dav_blk
template <int NC, bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dk_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                                   const float* __restrict__ Wk, const float* __restrict__ dav,
                                                                   float* __restrict__ dk, AttnDims d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* kks = smem + d.FP * d.XSS;
  float* qs = kks + d.FP * kRS;
  float* davs = qs + d.FP * kRS;
  const int b = blockIdx.x / d.H, h = blockIdx.x - b * d.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane & 15, g = lane >> 4;
  stage_x(x + (long)b * d.F * d.K, xs, d);
  const float* davh = dav + ((long)h * d.B + b) * d.F * d.A;
  for (int idx = threadIdx.x; idx < d.FP * 16; idx += kAttnThreads) {
    const int f = idx >> 4, aa = idx & 15;
    davs[f * kRS + aa] = (f < d.F && aa < d.A) ? davh[(long)f * d.A + aa] : 0.f;
  }
  float wq[NC][4], wk[NC][4];
  load_w<NC>(Wq, h, d, lane, wq);
  load_w<NC>(Wk, h, d, lane, wk);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) wq[c][s] *= scale;  // q pre-scaled: scores = (scale q) . k
  __syncthreads();
  project_all<NC, F16>(xs, kks, wk, d, wave, lane);
  project_all<NC, F16>(xs, qs, wq, d, wave, lane);
  __syncthreads();
  float* dkh = dk + ((long)h * d.B + b) * d.F * d.A;

  for (int t = wave; t < d.nblk; t += 4) {
    const float4 kB = lds_row4(kks, 16 * t + a, g);  // B operand: lane (j = f' = lane&15, k = g): k[f'][4g+s]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int blk = 0; blk < d.nblk; ++blk) {
      const float4 qA = lds_row4(qs, 16 * blk + a, g);     // A: (scale q)[f][4g+s]
      const float4 dA = lds_row4(davs, 16 * blk + a, g);   // A: dav[f][4g+s]
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, ds = {0.f, 0.f, 0.f, 0.f};
      sc = mma4<F16>(qA.x, qA.y, qA.z, qA.w, kB.x, kB.y, kB.z, kB.w, sc);
      ds = mma4<F16>(dA.x, dA.y, dA.z, dA.w, kB.x, kB.y, kB.z, kB.w, ds);
      // accumulators: row <-> f = 16 blk + 4g + r, col <-> f' = lane&15
      const float* qb = qs + (16 * blk + 4 * g) * kRS + a;
      const float* db = davs + (16 * blk + 4 * g) * kRS + a;
      float sgv[4], dpre[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sgv[s] = sigmoidf_fast(sc[s]);
        // qs holds scale*q: dpre (without the scale factor) times (scale q) == (dpre with scale) times q
        dpre[s] = ds[s] * sgv[s] * (1.f - sgv[s]);
      }
      acc = mma4<F16>(dpre[0], dpre[1], dpre[2], dpre[3], qb[0], qb[kRS], qb[2 * kRS], qb[3 * kRS], acc);
      acc = mma4<F16>(sgv[0], sgv[1], sgv[2], sgv[3], db[0], db[kRS], db[2 * kRS], db[3 * kRS], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r;
      if (f < d.F && a < d.A) dkh[(long)f * d.A + a] = acc[r];
    }
  }
}

// ================================================================================================= backward: projections
// rows R = B*F; D_j[row][h][a] for j in {dq, dk, dres};  dx[row][k] = sum_{j,h,a} D_j W_j[k][h][a];
// dW_j[k][h][a] = sum_row x[row][k] D_j[row][h][a]  (workgroup partials over a chunk of rows, reduced afterwards).
// One workgroup processes `rows_per_block` rows in tiles of 32 rows staged in LDS.
constexpr int kProjTile = 32;

__global__ __launch_bounds__(kAttnThreads) void attn_bwd_proj_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                                     const float* __restrict__ Wk, const float* __restrict__ Wr,
                                                                     const float* __restrict__ dq, const float* __restrict__ dk,
                                                                     const float* __restrict__ dr, float* __restrict__ dx,
                                                                     float* __restrict__ wpart /* [blocks][NJ][K][HA] */,
                                                                     AttnDims d, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HA = d.H * d.A;
  const int NJ = dr != nullptr ? 3 : 2;
  const int DW = NJ * HA;               // concatenated gradient width
  const int DS = DW + 1;                // LDS row stride (odd)
  const int KS = d.K + 1;
  float* Wc = smem;                     // [K][DW]   concatenated weights  W_j[k][h*A+a]
  float* Dt = Wc + d.K * DW;            // [tile][DS]
  float* Xt = Dt + kProjTile * DS;      // [tile][KS]
  const long R = (long)d.B * d.F;
  const long row_lo = (long)blockIdx.x * rows_per_block;
  const long row_hi = min(R, row_lo + rows_per_block);
  const float* Wj[3] = {Wq, Wk, Wr};
  const float* Dj[3] = {dq, dk, dr};
  for (int idx = threadIdx.x; idx < d.K * DW; idx += kAttnThreads) {
    const int k = idx / DW, c = idx - k * DW;
    const int j = c / HA, ha = c - j * HA;
    Wc[idx] = Wj[j][(long)k * HA + ha];
  }
  // dW accumulators: thread <-> (k = tid % K?, column group).  Outputs K*DW, dealt round-robin: o = tid + 256*u
  constexpr int kMaxOut = 48;           // K*DW <= 64*768 would be too many; host restricts K*DW <= 256*kMaxOut
  float acc[kMaxOut];
#pragma unroll
  for (int u = 0; u < kMaxOut; ++u) acc[u] = 0.f;
  const int nout = d.K * DW;
  __syncthreads();

  for (long r0 = row_lo; r0 < row_hi; r0 += kProjTile) {
    const int nr = (int)min((long)kProjTile, row_hi - r0);
    // stage D tile: D_j[h][row][a] -> Dt[row][j*HA + h*A + a]
    for (int idx = threadIdx.x; idx < kProjTile * DW; idx += kAttnThreads) {
      const int row = idx / DW, c = idx - row * DW;
      float v = 0.f;
      if (row < nr) {
        const int j = c / HA, ha = c - j * HA;
        const int hh = ha / d.A, aa = ha - hh * d.A;
        v = Dj[j][((long)hh * R + (r0 + row)) * d.A + aa];
      }
      Dt[row * DS + c] = v;
    }
    for (int idx = threadIdx.x; idx < kProjTile * d.K; idx += kAttnThreads) {
      const int row = idx / d.K, k = idx - row * d.K;
      Xt[row * KS + k] = row < nr ? x[(r0 + row) * d.K + k] : 0.f;
    }
    __syncthreads();
    // dx[row][k] = sum_c Dt[row][c] Wc[k][c]
    for (int idx = threadIdx.x; idx < kProjTile * d.K; idx += kAttnThreads) {
      const int row = idx / d.K, k = idx - row * d.K;
      if (row < nr) {
        const float* dp = Dt + row * DS;
        const float* wp = Wc + k * DW;
        float t = 0.f;
        for (int c = 0; c < DW; ++c) t = fmaf(dp[c], wp[c], t);
        dx[(r0 + row) * d.K + k] = t;
      }
    }
    // dW[k][c] += sum_row Xt[row][k] Dt[row][c]
#pragma unroll
    for (int u = 0; u < kMaxOut; ++u) {
      const int o = threadIdx.x + u * kAttnThreads;
      if (o < nout) {
        const int k = o / DW, c = o - k * DW;
        float t = acc[u];
        for (int row = 0; row < kProjTile; ++row) t = fmaf(Xt[row * KS + k], Dt[row * DS + c], t);
        acc[u] = t;
      }
    }
    __syncthreads();
  }
  float* wp = wpart + (long)blockIdx.x * nout;
#pragma unroll
  for (int u = 0; u < kMaxOut; ++u) {
    const int o = threadIdx.x + u * kAttnThreads;
    if (o < nout) wp[o] = acc[u];
  }
}

// MFMA form of the projection backward (used when NJ*H*NC <= kProjMaxSeg).  A wave walks 16-row tiles of the
// flattened rows R = B*F; per (j,h) segment it issues
//   dx_tile [16 x K]  += D_seg [16 x A] W_seg^T [A x K]     (A operand: D rows, 16-byte loads; B: W_seg from LDS)
//   dW_seg  [K x A]   += x_tile^T [K x 16] D_seg [16 x A]   (A operand: x columns; B: D columns)
// and keeps the dW accumulators in registers across all its tiles (per-wave partials, reduced afterwards).
constexpr int kProjMaxSeg = 24;

template <int NC, bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_proj3_kernel(const float* __restrict__ x, const float* __restrict__ Wq,
                                                                      const float* __restrict__ Wk, const float* __restrict__ Wr,
                                                                      const float* __restrict__ dq, const float* __restrict__ dk,
                                                                      const float* __restrict__ dr, float* __restrict__ dx,
                                                                      float* __restrict__ wpart, AttnDims d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // Wl[NS][16*NC][kRS]
  const int NJ = dr != nullptr ? 3 : 2;
  const int NS = NJ * d.H, HA = d.H * d.A, DW = NJ * HA;
  const long R = (long)d.B * d.F;
  const float* Wj[3] = {Wq, Wk, Wr};
  const float* Dj[3] = {dq, dk, dr};
  for (int idx = threadIdx.x; idx < NS * 16 * NC * 16; idx += kAttnThreads) {
    const int a = idx & 15, k = (idx >> 4) % (16 * NC), sg = idx / (16 * NC * 16);
    const int j = sg / d.H, h = sg - j * d.H;
    smem[(sg * 16 * NC + k) * kRS + a] = (k < d.K && a < d.A) ? Wj[j][((long)k * d.H + h) * d.A + a] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  float* tsc = smem + NS * 16 * NC * kRS + wave * (2 * 16 * kRS);   // per-wave transpose tiles (behind the staged weights)
  f32x4 accw[kProjMaxSeg];
#pragma unroll
  for (int u = 0; u < kProjMaxSeg; ++u) accw[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long ntiles = (R + 15) / 16;
  const int wglob = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  for (long tile = wglob; tile < ntiles; tile += nw) {
    const long r0 = tile * 16;
    float xa[NC][4];
#pragma unroll
    for (int kt = 0; kt < NC; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const long row = r0 + 4 * g + s2;
        const int k = 16 * kt + li;
        xa[kt][s2] = (row < R && k < d.K) ? x[row * d.K + k] : 0.f;
      }
    f32x4 accx[NC];
#pragma unroll
    for (int kt = 0; kt < NC; ++kt) accx[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long rowi = r0 + li;
#pragma unroll
    for (int sgk = 0; sgk < kProjMaxSeg / NC; ++sgk) {
      if (sgk < NS) {
        const int j = sgk / d.H, h = sgk - j * d.H;
        const float* Dp = Dj[j] + (long)h * R * d.A;
        // the 16x16 block D[r0..r0+15][0..15] is needed in both orientations (dv: lane's row li, columns 4g..4g+3;
        // dc: rows 4g..4g+3, lane's column li).  It is read from memory once (dv) and turned through a per-wave LDS
        // tile for dc: reading it twice made this kernel move 1.26 GB per call at one wave per SIMD.
        float dv[4], dc[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const int a = 4 * g + s2;
          dv[s2] = (rowi < R && a < d.A) ? Dp[rowi * d.A + a] : 0.f;
        }
        float* tw = tsc + (sgk & 1) * (16 * kRS);   // two tiles: the next segment's write cannot hit this one's reads
        *reinterpret_cast<float4*>(tw + li * kRS + 4 * g) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) dc[s2] = tw[(4 * g + s2) * kRS + li];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < NC; ++kt) {
          const float4 wb = *reinterpret_cast<const float4*>(smem + (sgk * 16 * NC + 16 * kt + li) * kRS + 4 * g);
          accx[kt] = mma4<F16>(dv[0], dv[1], dv[2], dv[3], wb.x, wb.y, wb.z, wb.w, accx[kt]);
          f32x4& aw = accw[sgk * NC + kt];
          aw = mma4<F16>(xa[kt][0], xa[kt][1], xa[kt][2], xa[kt][3], dc[0], dc[1], dc[2], dc[3], aw);
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < NC; ++kt)
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        const long row = r0 + 4 * g + r2;
        const int k = 16 * kt + li;
        if (row < R && k < d.K) dx[row * d.K + k] = accx[kt][r2];
      }
  }
  float* wp = wpart + (long)wglob * d.K * DW;
#pragma unroll
  for (int sgk = 0; sgk < kProjMaxSeg / NC; ++sgk) {
    if (sgk < NS) {
#pragma unroll
      for (int kt = 0; kt < NC; ++kt)
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
          const int k = 16 * kt + 4 * g + r2;
          if (k < d.K && li < d.A) wp[(long)k * DW + sgk * d.A + li] = accw[sgk * NC + kt][r2];
        }
    }
  }
}

// out[i] = sum_p part[p*n + i]  (fixed order): 64 outputs per workgroup, the 4 waves take every 4th partial
__global__ __launch_bounds__(256) void attn_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int parts) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < n) {
    int p = wave;
    for (; p + 12 < parts; p += 16) {
      t0 += part[(long)p * n + i];
      t1 += part[(long)(p + 4) * n + i];
      t2 += part[(long)(p + 8) * n + i];
      t3 += part[(long)(p + 12) * n + i];
    }
    for (; p < parts; p += 4) t0 += part[(long)p * n + i];
  }
  red[wave][lane] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (wave == 0 && i < n) out[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// splits the reduced [NJ][K][HA]-as-[K][DW] buffer into dWq, dWk, dWr ([K][H][A] each)
__global__ __launch_bounds__(256) void attn_split_dw_kernel(const float* __restrict__ red, float* __restrict__ dWq,
                                                            float* __restrict__ dWk, float* __restrict__ dWr, int K, int HA, int NJ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int DW = NJ * HA;
  if (i >= K * DW) return;
  const int k = i / DW, c = i - k * DW;
  const int j = c / HA, ha = c - j * HA;
  float* dst = j == 0 ? dWq : (j == 1 ? dWk : dWr);
  dst[(long)k * HA + ha] = red[i];
}

// dgamma/dbeta: sum the [blocks][2][16] partials; one workgroup per output, strided partial sums + fixed-order tree
__global__ __launch_bounds__(256) void attn_reduce_gb_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int blocks, int A) {
  __shared__ float red[256];
  const int which = blockIdx.x >> 4, a = blockIdx.x & 15;
  float t = 0.f;
  for (int p = threadIdx.x; p < blocks; p += 256) t += part[((long)p * 2 + which) * 16 + a];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0 && a < A) (which == 0 ? dgamma : dbeta)[a] = red[0];
}

// ------------------------------------------------------------------------------------------------- host
static int make_dims(const char* fn, int B, int F, int K, int H, int A, AttnDims& d) {
  if (B < 0 || F < 1 || K < 1 || H < 1 || A < 1) return fail(FIL_ERR_ARG, "%s: bad shape B=%d F=%d K=%d H=%d A=%d", fn, B, F, K, H, A);
  if (K > 16 * kMaxNC) return fail(FIL_ERR_UNSUPPORTED, "%s: K=%d > %d", fn, K, 16 * kMaxNC);
  if (A > 16) return fail(FIL_ERR_UNSUPPORTED, "%s: attention_dim A=%d > 16", fn, A);
  if (F > 512) return fail(FIL_ERR_UNSUPPORTED, "%s: F=%d > 512 fields", fn, F);
  d.B = B; d.F = F; d.K = K; d.H = H; d.A = A;
  d.nblk = cdiv(F, 16);
  d.FP = 16 * d.nblk;
  d.NC = cdiv(K, 16);
  d.XSS = 16 * d.NC + 4;
  return FIL_OK;
}

static int proj_rows_per_block(const AttnDims& d) {
  const long R = (long)d.B * d.F;
  long rpb = cdiv((int)std::min<long>(R, 1L << 30), 1024);
  rpb = std::max<long>(kProjTile, (rpb + kProjTile - 1) / kProjTile * kProjTile);
  return (int)rpb;
}
static int proj_blocks(const AttnDims& d) { return (int)(((long)d.B * d.F + proj_rows_per_block(d) - 1) / proj_rows_per_block(d)); }
static bool proj_mfma_ok(const AttnDims& d) { return 3 * d.H * d.NC <= kProjMaxSeg; }
static int proj3_blocks(const AttnDims& d) {
  const char* e = getenv("FIL_ATTN_PROJ_BLOCKS");   // tuning knob (results identical up to the partial-sum order)
  const long cap = e != nullptr && atoi(e) > 0 ? atoi(e) : 512;
  return (int)std::max<long>(1, std::min<long>(((long)d.B * d.F + 63) / 64, cap));
}
static int proj_parts(const AttnDims& d) { return proj_mfma_ok(d) ? 4 * proj3_blocks(d) : proj_blocks(d); }

static size_t attn_bwd_ws(const AttnDims& d) {
  const size_t act = align_up((size_t)d.H * d.B * d.F * d.A * sizeof(float), 256);
  size_t t = 4 * act;                                                                  // dav, dres, dq, dk
  t += align_up((size_t)d.B * d.H * 2 * 16 * sizeof(float), 256);                       // dgamma/dbeta partials
  t += align_up((size_t)(proj_parts(d) + 1) * 3 * d.K * d.H * d.A * sizeof(float), 256);  // dW partials + reduced
  return t;
}

template <typename KernelT>
static void allow_lds_attn(KernelT kernel, size_t sh) {
  if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
}

// CALL(NC, F16) for the runtime (NC, precision) pair; `f16` must be in scope
#define FIL_ATTN_NC(NCV, CALL)                                               \
  switch (NCV) {                                                             \
    case 1: { if (f16) { CALL(1, true); } else { CALL(1, false); } } break;  \
    case 2: { if (f16) { CALL(2, true); } else { CALL(2, false); } } break;  \
    case 3: { if (f16) { CALL(3, true); } else { CALL(3, false); } } break;  \
    case 4: { if (f16) { CALL(4, true); } else { CALL(4, false); } } break;  \
  }

}  // namespace fil

using namespace fil;

extern "C" size_t fil_attn_fwd_workspace_bytes(int, int, int, int, int) { return 0; }

extern "C" size_t fil_attn_bwd_workspace_bytes(int B, int F, int K, int H, int A) {
  AttnDims d;
  if (make_dims("fil_attn_bwd_workspace_bytes", B, F, K, H, A, d) != FIL_OK || B == 0) return 0;
  return attn_bwd_ws(d);
}

extern "C" int fil_attn_fwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                            const float* beta, float* y, float* res_out, float* av_out, int B, int F, int K, int H, int A,
                            float scale, float eps, int fuse_relu, int precision, void* workspace, size_t workspace_bytes,
                            void* stream) {
  (void)workspace; (void)workspace_bytes;
  AttnDims d;
  int rc = make_dims("fil_attn_fwd", B, F, K, H, A, d);
  if (rc != FIL_OK) return rc;
  if (precision != FIL_PREC_F32 && precision != FIL_PREC_F16_MFMA) return fail(FIL_ERR_ARG, "fil_attn_fwd: precision=%d", precision);
  const bool f16 = precision == FIL_PREC_F16_MFMA;
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(x && Wq && Wk && y);
  FIL_CHECK_ARG((gamma == nullptr) == (beta == nullptr));
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = ((size_t)d.FP * d.XSS + (size_t)d.FP * kRS) * sizeof(float);
  if (sh > 160 * 1024) return fail(FIL_ERR_UNSUPPORTED, "fil_attn_fwd: F=%d K=%d needs %zu bytes of LDS (> 160 KiB)", F, K, sh);
  const dim3 grid(B * H);
  // flops: projections 2*F*K*A*(2 or 3) + scores and weighted sum 2*2*F*F*A, per (b,h)
  ProfScope ps("attn_fwd", st, (double)B * H * (2.0 * F * K * A * (Wr ? 3 : 2) + 4.0 * F * (double)F * A));
#define CALL_FWD(N, P)                                                                                                     \
  allow_lds_attn(attn_fwd_kernel<N, P>, sh);                                                                               \
  hipLaunchKernelGGL((attn_fwd_kernel<N, P>), grid, dim3(kAttnThreads), sh, st, x, Wq, Wk, Wr, gamma, beta, y, res_out, \
                     av_out, d, scale, eps, fuse_relu)
  FIL_ATTN_NC(d.NC, CALL_FWD)
#undef CALL_FWD
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_attn_bwd(const float* x, const float* Wq, const float* Wk, const float* Wr, const float* gamma,
                            const float* beta, const float* dy, const float* dres_in, const float* y_saved,
                            const float* av_saved, float* dx, float* dWq, float* dWk, float* dWr, float* dgamma, float* dbeta,
                            int B, int F, int K, int H, int A, float scale, float eps, int fuse_relu, int precision,
                            void* workspace, size_t workspace_bytes, void* stream) {
  AttnDims d;
  int rc = make_dims("fil_attn_bwd", B, F, K, H, A, d);
  if (rc != FIL_OK) return rc;
  if (precision != FIL_PREC_F32 && precision != FIL_PREC_F16_MFMA) return fail(FIL_ERR_ARG, "fil_attn_bwd: precision=%d", precision);
  const bool f16 = precision == FIL_PREC_F16_MFMA;
  FIL_CHECK_ARG(Wq && Wk && dWq && dWk);
  FIL_CHECK_ARG((gamma == nullptr) == (beta == nullptr));
  FIL_CHECK_ARG(gamma == nullptr || (dgamma && dbeta));
  FIL_CHECK_ARG(Wr == nullptr || dWr != nullptr);
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
  if (workspace == nullptr || workspace_bytes < attn_bwd_ws(d))
    return fail(FIL_ERR_WORKSPACE, "fil_attn_bwd: workspace %zu < %zu bytes", workspace_bytes, attn_bwd_ws(d));
  // unfused mode: the residual branch's gradient arrives separately (dres_in); fused: produced by the pre kernel
  const bool has_res = Wr != nullptr;
  if (!fuse_relu && has_res && dres_in == nullptr) return fail(FIL_ERR_ARG, "fil_attn_bwd: dres is required when fuse_relu == 0 and Wr != NULL");
  const int NJ = has_res ? 3 : 2;
  if ((long)K * NJ * H * A > 256L * 48) return fail(FIL_ERR_UNSUPPORTED, "fil_attn_bwd: K*%d*H*A = %ld > 12288", NJ, (long)K * NJ * H * A);
  Carver ws(workspace);
  const size_t nact = (size_t)H * B * F * A;
  float* dav = ws.take<float>(nact);
  float* dres = ws.take<float>(nact);
  float* dq = ws.take<float>(nact);
  float* dk = ws.take<float>(nact);
  float* gb_part = ws.take<float>((size_t)B * H * 2 * 16);
  const int pblocks = proj_parts(d);
  const int nout = K * NJ * H * A;
  float* wpart = ws.take<float>((size_t)(pblocks + 1) * 3 * K * H * A);
  float* wred = wpart + (size_t)pblocks * nout;

  const dim3 grid(B * H);
  const size_t sh_base = ((size_t)d.FP * d.XSS + (size_t)d.FP * kRS) * sizeof(float);
  if (sh_base + 2 * (size_t)d.FP * kRS * sizeof(float) > 160 * 1024)
    return fail(FIL_ERR_UNSUPPORTED, "fil_attn_bwd: F=%d K=%d needs more than 160 KiB of LDS", F, K);
  const double core = (double)B * H * 4.0 * F * (double)F * A;  // one score + one weighted-sum pass
  // saved-av path: LayerNorm needs av; the fused ReLU mask needs y.  Anything missing -> recompute (original kernel).
  const bool saved_path = (gamma == nullptr || av_saved != nullptr) && (!fuse_relu || y_saved != nullptr);
  int gb_blocks = B * H;
  if (saved_path) {
    const long rows = (long)H * B * F;
    gb_blocks = (int)std::min<long>((rows + 15) / 16, (long)B * H);   // gb_part holds B*H blocks of partials
    ProfScope ps("attn_bwd_pre", st, (double)rows * A * 5 * sizeof(float));
    hipLaunchKernelGGL(attn_bwd_pre_saved_kernel, dim3(gb_blocks), dim3(kAttnThreads), 0, st, av_saved, y_saved, gamma, dy, dav,
                       (fuse_relu && has_res) ? dres : nullptr, gb_part, rows, A, eps, fuse_relu);
  } else {
    const size_t sh = sh_base + 4 * 2 * 16 * sizeof(float);
    ProfScope ps("attn_bwd_pre", st, core);
#define CALL_PRE(N, P)                                                                                                      \
  allow_lds_attn(attn_bwd_pre_kernel<N, P>, sh);                                                                            \
  hipLaunchKernelGGL((attn_bwd_pre_kernel<N, P>), grid, dim3(kAttnThreads), sh, st, x, Wq, Wk, Wr, gamma, beta, dy, dav,    \
                     (fuse_relu && has_res) ? dres : nullptr, gb_part, d, scale, eps, fuse_relu)
    FIL_ATTN_NC(d.NC, CALL_PRE)
#undef CALL_PRE
  }
  FIL_CHECK_LAUNCH();
  if (gamma != nullptr) {
    hipLaunchKernelGGL(attn_reduce_gb_kernel, dim3(32), dim3(256), 0, st, gb_part, dgamma, dbeta, gb_blocks, A);
    FIL_CHECK_LAUNCH();
  }
  {
    ProfScope ps("attn_bwd_dq", st, core * 1.5);
#define CALL_DQ(N, P)                                                                                                       \
  allow_lds_attn(attn_bwd_dq_kernel<N, P>, sh_base);                                                                        \
  hipLaunchKernelGGL((attn_bwd_dq_kernel<N, P>), grid, dim3(kAttnThreads), sh_base, st, x, Wq, Wk, dav, dq, d, scale)
    FIL_ATTN_NC(d.NC, CALL_DQ)
#undef CALL_DQ
  }
  FIL_CHECK_LAUNCH();
  {
    const size_t sh = sh_base + 2 * (size_t)d.FP * kRS * sizeof(float);
    ProfScope ps("attn_bwd_dk", st, core * 2.0);
#define CALL_DK(N, P)                                                                                                       \
  allow_lds_attn(attn_bwd_dk_kernel<N, P>, sh);                                                                             \
  hipLaunchKernelGGL((attn_bwd_dk_kernel<N, P>), grid, dim3(kAttnThreads), sh, st, x, Wq, Wk, dav, dk, d, scale)
    FIL_ATTN_NC(d.NC, CALL_DK)
#undef CALL_DK
  }
  FIL_CHECK_LAUNCH();
  {
    const float* drsrc = has_res ? (fuse_relu ? dres : dres_in) : nullptr;
    const int DW = NJ * H * A;
    const size_t sh = ((size_t)K * DW + (size_t)kProjTile * (DW + 1) + (size_t)kProjTile * (K + 1)) * sizeof(float);
    if (!proj_mfma_ok(d) && sh > 150 * 1024) return fail(FIL_ERR_UNSUPPORTED, "fil_attn_bwd: projection tile needs %zu bytes of LDS", sh);
    ProfScope ps("attn_bwd_proj", st, (double)B * F * 4.0 * K * DW);
    if (proj_mfma_ok(d)) {
      const size_t sh3 = ((size_t)NJ * H * 16 * d.NC * kRS + 4 * 2 * 16 * kRS) * sizeof(float);
#define CALL_PROJ3(N, P)                                                                                                    \
  allow_lds_attn(attn_bwd_proj3_kernel<N, P>, sh3);                                                                         \
  hipLaunchKernelGGL((attn_bwd_proj3_kernel<N, P>), dim3(proj3_blocks(d)), dim3(kAttnThreads), sh3, st, x, Wq, Wk, Wr, dq, dk, drsrc, \
                     dx, wpart, d)
      FIL_ATTN_NC(d.NC, CALL_PROJ3)
#undef CALL_PROJ3
    } else {
      allow_lds_attn(attn_bwd_proj_kernel, sh);
      hipLaunchKernelGGL(attn_bwd_proj_kernel, dim3(pblocks), dim3(kAttnThreads), sh, st, x, Wq, Wk, Wr, dq, dk, drsrc, dx, wpart, d,
                         proj_rows_per_block(d));
    }
    FIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_reduce_kernel, dim3(cdiv(nout, 64)), dim3(256), 0, st, wpart, wred, nout, pblocks);
    FIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_split_dw_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, st, wred, dWq, dWk, dWr, K, H * A, NJ);
    FIL_CHECK_LAUNCH();
  }
  return FIL_OK;
}
