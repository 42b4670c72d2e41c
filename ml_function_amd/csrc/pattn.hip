// ProductAttentionLayer on explicit [q, k, v] (+ mask) for gfx950 -- the stand-alone form of the attention core
// (behavior_layer.py:292-311): out = sigmoid(scale * q k^T [+ mask * (-1e5)]) v, the "softmax" being a sigmoid (:286,308).
//
// This is the layer's general entry point (separate q/k/v, any mask); AutoInt's hot path is the fused kernel pair in
// attn.hip (projections + attention + LayerNorm + residual in one pass).  Exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), the
// F x F scores never leave registers.  mask_mod == 2 (additive -1e5 mask) is applied in the kernel; mask_mod == 1 (the
// scores right-multiplied by a mask matrix, :300-302) is algebra on k -- (q k^T) M = q (M^T k)^T -- done by the caller.
//
//   pattn_fwd_kernel   workgroup = one item n (e.g. one (head, sample)); k and v staged in LDS; a wave owns 16-query blocks:
//                      S^T[key][query] = sigmoid(.) with the key on (lane>>4, reg) -> it IS the B operand of
//                      out^T[av][query] += v^T S^T; out^T leaves as [query on the lane][4 consecutive av] (16-byte stores).
//   pattn_bwd_q_kernel same orientation: dS^T = v dout^T, dP = dS S (1-S) scale, dq^T[a][query] += k^T dP^T.
//   pattn_bwd_kv_kernel a wave owns 16-key blocks and walks the queries (q, dout staged in LDS):
//                      S[query][key], dS = dout v^T;  dk^T[a][key] += q^T dP,  dv^T[av][key] += dout^T S.
// The backward evaluates the scores twice (once per orientation); every sum has a fixed order (bit-identical repeats).
#include "common.h"

namespace fil {

constexpr int kPRS = 20;     // LDS row stride of [rows][16] fp32 tiles
constexpr int kPMaxC = 4;    // A, Av <= 64 (chunks of 16)

__device__ __forceinline__ f32x4 pmma(f32x4 a, f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}

// img[ch][row][kPRS]: element (row, 16 ch + col) of a [rows x W] matrix, zero padded to [RP x 16 NCH]
__device__ __forceinline__ void pstage(const float* __restrict__ src, float* img, int rows, int RP, int W, int NCH, int nthreads) {
  for (int idx = threadIdx.x; idx < NCH * RP * 16; idx += nthreads) {
    const int col = idx & 15, row = (idx >> 4) % RP, ch = idx / (16 * RP);
    const int w = 16 * ch + col;
    img[(ch * RP + row) * kPRS + col] = (row < rows && w < W) ? src[(long)row * W + w] : 0.f;
  }
}
// row fragment: lane (g,c) -> img[row0 + c][4g .. 4g+3] of chunk ch
__device__ __forceinline__ f32x4 prow(const float* img, int RP, int ch, int row0, int lane) {
  return *reinterpret_cast<const f32x4*>(img + (ch * RP + row0 + (lane & 15)) * kPRS + 4 * (lane >> 4));
}
// column fragment: lane (g,c) -> img[row0 + 4g + s][c] of chunk ch, s = 0..3
__device__ __forceinline__ f32x4 pcol(const float* img, int RP, int ch, int row0, int lane) {
  const float* p = img + (ch * RP + row0 + 4 * (lane >> 4)) * kPRS + (lane & 15);
  return f32x4{p[0], p[kPRS], p[2 * kPRS], p[3 * kPRS]};
}
// row fragment straight from global: rows of a [R x W] matrix, lane (g,c) -> src[row0 + c][16 ch + 4g + s]
__device__ __forceinline__ f32x4 grow(const float* __restrict__ src, int R, int W, int ch, int row0, int lane) {
  const int r = row0 + (lane & 15), w0 = 16 * ch + 4 * (lane >> 4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < R) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (w0 + s < W) v[s] = src[(long)r * W + w0 + s];
  }
  return v;
}
__device__ __forceinline__ void gstore_rowT(float* __restrict__ dst, int R, int W, int ch, int row0, int lane, const f32x4& v) {
  // v is a transposed product tile D[w 4g+r][row c]: lane (g,c) owns dst[row0 + c][16 ch + 4g + r]
  const int r = row0 + (lane & 15), w0 = 16 * ch + 4 * (lane >> 4);
  if (r < R) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (w0 + s < W) dst[(long)r * W + w0 + s] = v[s];
  }
}

struct PDims {
  int N, Fq, Fk, A, Av;
  int nq, nk;        // ceil(F/16)
  int QP, KP;        // padded rows
  int NA, NV;        // chunks of A, Av
  int mask_period;   // mask item = n % mask_period (0: no mask)
};

__device__ __forceinline__ float psigmoid(float t) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t)); }

// additive mask term of the S^T tile (rows = keys 16j + 4g + r, column = query 16i + c), already times -log2(e):
// scores are kept as t = -log2(e) * (scale q.k + mask * (-1e5))
__device__ __forceinline__ f32x4 mask_T(const float* __restrict__ m, const PDims& d, int i, int j, int lane) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int q = 16 * i + (lane & 15), k0 = 16 * j + 4 * (lane >> 4);
  if (m != nullptr && q < d.Fq) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (k0 + s < d.Fk) v[s] = m[(long)q * d.Fk + k0 + s] * (1e5f * 1.4426950408889634f);
  }
  return v;
}
// the same for the S tile (rows = queries 16i + 4g + r, column = key 16j + c)
__device__ __forceinline__ f32x4 mask_D(const float* __restrict__ m, const PDims& d, int i, int j, int lane) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int k = 16 * j + (lane & 15), q0 = 16 * i + 4 * (lane >> 4);
  if (m != nullptr && k < d.Fk) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (q0 + s < d.Fq) v[s] = m[(long)(q0 + s) * d.Fk + k] * (1e5f * 1.4426950408889634f);
  }
  return v;
}

// MODE 0: forward (out).  MODE 1: backward dq (needs dout).
template <int MODE>
__global__ __launch_bounds__(256) void pattn_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                      const float* __restrict__ mask, const float* __restrict__ dout,
                                                      float* __restrict__ out /* out or dq */, PDims d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* kimg = psm;                                  // [NA][KP][kPRS]
  float* vimg = psm + d.NA * d.KP * kPRS;             // [NV][KP][kPRS]
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* qn = q + (long)n * d.Fq * d.A;
  const float* mn = mask != nullptr ? mask + (long)(n % d.mask_period) * d.Fq * d.Fk : nullptr;
  pstage(k + (long)n * d.Fk * d.A, kimg, d.Fk, d.KP, d.A, d.NA, 256);
  pstage(v + (long)n * d.Fk * d.Av, vimg, d.Fk, d.KP, d.Av, d.NV, 256);
  __syncthreads();
  const float qs = -scale * 1.4426950408889634f;
  for (int i = wave; i < d.nq; i += 4) {
    f32x4 qf[kPMaxC], dof[kPMaxC], acc[kPMaxC];
#pragma unroll
    for (int ca = 0; ca < kPMaxC; ++ca) {
      qf[ca] = ca < d.NA ? grow(qn, d.Fq, d.A, ca, 16 * i, lane) * qs : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[ca] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (MODE == 1) {
#pragma unroll
      for (int cv = 0; cv < kPMaxC; ++cv)
        dof[cv] = cv < d.NV ? grow(dout + (long)n * d.Fq * d.Av, d.Fq, d.Av, cv, 16 * i, lane) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int j = 0; j < d.nk; ++j) {
      f32x4 sc = mask_T(mn, d, i, j, lane);                        // S^T tile: [key 4g+r][query c]
#pragma unroll
      for (int ca = 0; ca < kPMaxC; ++ca)
        if (ca < d.NA) sc = pmma(prow(kimg, d.KP, ca, 16 * j, lane), qf[ca], sc);
      f32x4 sg;
#pragma unroll
      for (int r = 0; r < 4; ++r) sg[r] = psigmoid(sc[r]);
      if (MODE == 0) {
#pragma unroll
        for (int cv = 0; cv < kPMaxC; ++cv)                         // out^T[av][query] += v^T S^T
          if (cv < d.NV) acc[cv] = pmma(pcol(vimg, d.KP, cv, 16 * j, lane), sg, acc[cv]);
      } else {
        f32x4 ds = {0.f, 0.f, 0.f, 0.f};                            // dS^T[key][query] = v dout^T
#pragma unroll
        for (int cv = 0; cv < kPMaxC; ++cv)
          if (cv < d.NV) ds = pmma(prow(vimg, d.KP, cv, 16 * j, lane), dof[cv], ds);
        f32x4 dp;
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[r] = ds[r] * sg[r] * (1.f - sg[r]) * scale;
#pragma unroll
        for (int ca = 0; ca < kPMaxC; ++ca)                         // dq^T[a][query] += k^T dP^T
          if (ca < d.NA) acc[ca] = pmma(pcol(kimg, d.KP, ca, 16 * j, lane), dp, acc[ca]);
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int cv = 0; cv < kPMaxC; ++cv)
        if (cv < d.NV) gstore_rowT(out + (long)n * d.Fq * d.Av, d.Fq, d.Av, cv, 16 * i, lane, acc[cv]);
    } else {
#pragma unroll
      for (int ca = 0; ca < kPMaxC; ++ca)
        if (ca < d.NA) gstore_rowT(out + (long)n * d.Fq * d.A, d.Fq, d.A, ca, 16 * i, lane, acc[ca]);
    }
  }
}

__global__ __launch_bounds__(256) void pattn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ mask,
                                                           const float* __restrict__ dout, float* __restrict__ dk, float* __restrict__ dv,
                                                           PDims d, float scale) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* qimg = psm;                                  // [NA][QP][kPRS]
  float* oimg = psm + d.NA * d.QP * kPRS;             // [NV][QP][kPRS]  (dout)
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* kn = k + (long)n * d.Fk * d.A;
  const float* vn = v + (long)n * d.Fk * d.Av;
  const float* mn = mask != nullptr ? mask + (long)(n % d.mask_period) * d.Fq * d.Fk : nullptr;
  pstage(q + (long)n * d.Fq * d.A, qimg, d.Fq, d.QP, d.A, d.NA, 256);
  pstage(dout + (long)n * d.Fq * d.Av, oimg, d.Fq, d.QP, d.Av, d.NV, 256);
  __syncthreads();
  const float qs = -scale * 1.4426950408889634f;
  for (int j = wave; j < d.nk; j += 4) {
    f32x4 kf[kPMaxC], vf[kPMaxC], dka[kPMaxC], dva[kPMaxC];
#pragma unroll
    for (int c = 0; c < kPMaxC; ++c) {
      kf[c] = c < d.NA ? grow(kn, d.Fk, d.A, c, 16 * j, lane) : f32x4{0.f, 0.f, 0.f, 0.f};
      vf[c] = c < d.NV ? grow(vn, d.Fk, d.Av, c, 16 * j, lane) : f32x4{0.f, 0.f, 0.f, 0.f};
      dka[c] = dva[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < d.nq; ++i) {
      f32x4 sc = mask_D(mn, d, i, j, lane);                         // S tile: [query 4g+r][key c]
#pragma unroll
      for (int ca = 0; ca < kPMaxC; ++ca)
        if (ca < d.NA) sc = pmma(prow(qimg, d.QP, ca, 16 * i, lane) * qs, kf[ca], sc);
      f32x4 ds = {0.f, 0.f, 0.f, 0.f};                              // dS[query][key] = dout v^T
#pragma unroll
      for (int cv = 0; cv < kPMaxC; ++cv)
        if (cv < d.NV) ds = pmma(prow(oimg, d.QP, cv, 16 * i, lane), vf[cv], ds);
      f32x4 sg, dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sg[r] = psigmoid(sc[r]);
        dp[r] = ds[r] * sg[r] * (1.f - sg[r]) * scale;
      }
#pragma unroll
      for (int ca = 0; ca < kPMaxC; ++ca)                           // dk^T[a][key] += q^T dP
        if (ca < d.NA) dka[ca] = pmma(pcol(qimg, d.QP, ca, 16 * i, lane), dp, dka[ca]);
#pragma unroll
      for (int cv = 0; cv < kPMaxC; ++cv)                           // dv^T[av][key] += dout^T S
        if (cv < d.NV) dva[cv] = pmma(pcol(oimg, d.QP, cv, 16 * i, lane), sg, dva[cv]);
    }
#pragma unroll
    for (int c = 0; c < kPMaxC; ++c) {
      if (c < d.NA) gstore_rowT(dk + (long)n * d.Fk * d.A, d.Fk, d.A, c, 16 * j, lane, dka[c]);
      if (c < d.NV) gstore_rowT(dv + (long)n * d.Fk * d.Av, d.Fk, d.Av, c, 16 * j, lane, dva[c]);
    }
  }
}

static int pdims(const char* fn, int N, int Fq, int Fk, int A, int Av, int mask_period, const void* mask, PDims& d) {
  if (N < 0 || Fq < 1 || Fk < 1 || A < 1 || Av < 1) return fail(FIL_ERR_ARG, "%s: bad shape N=%d Fq=%d Fk=%d A=%d Av=%d", fn, N, Fq, Fk, A, Av);
  if (A > 16 * kPMaxC || Av > 16 * kPMaxC) return fail(FIL_ERR_UNSUPPORTED, "%s: A=%d / Av=%d > %d", fn, A, Av, 16 * kPMaxC);
  if (mask != nullptr && (mask_period < 1 || (N > 0 && N % mask_period != 0)))
    return fail(FIL_ERR_ARG, "%s: mask_period=%d must divide N=%d", fn, mask_period, N);
  d.N = N; d.Fq = Fq; d.Fk = Fk; d.A = A; d.Av = Av;
  d.nq = cdiv(Fq, 16); d.nk = cdiv(Fk, 16);
  d.QP = 16 * d.nq; d.KP = 16 * d.nk;
  d.NA = cdiv(A, 16); d.NV = cdiv(Av, 16);
  d.mask_period = mask != nullptr ? mask_period : 0;
  return FIL_OK;
}

template <typename KernelT>
static int pallow(KernelT kernel, size_t sh) {
  if (sh > 160 * 1024) return FIL_ERR_UNSUPPORTED;
  if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  return FIL_OK;
}

}  // namespace fil

using namespace fil;

extern "C" int fil_pattn_fwd(const float* q, const float* k, const float* v, const float* mask, float* out, int N, int Fq, int Fk,
                             int A, int Av, float scale, int mask_period, void* stream) {
  PDims d;
  int rc = pdims("fil_pattn_fwd", N, Fq, Fk, A, Av, mask_period, mask, d);
  if (rc != FIL_OK) return rc;
  if (N == 0) return FIL_OK;
  FIL_CHECK_ARG(q && k && v && out);
  const size_t sh = (size_t)(d.NA + d.NV) * d.KP * kPRS * sizeof(float);
  if (pallow(pattn_q_kernel<0>, sh) != FIL_OK) return fail(FIL_ERR_UNSUPPORTED, "fil_pattn_fwd: Fk=%d needs %zu bytes of LDS (> 160 KiB)", Fk, sh);
  ProfScope ps("pattn_fwd", (hipStream_t)stream, (double)N * 2.0 * Fq * (double)Fk * (A + Av));
  hipLaunchKernelGGL(pattn_q_kernel<0>, dim3(N), dim3(256), sh, (hipStream_t)stream, q, k, v, mask, nullptr, out, d, scale);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_pattn_bwd(const float* q, const float* k, const float* v, const float* mask, const float* dout, float* dq,
                             float* dk, float* dv, int N, int Fq, int Fk, int A, int Av, float scale, int mask_period, void* stream) {
  PDims d;
  int rc = pdims("fil_pattn_bwd", N, Fq, Fk, A, Av, mask_period, mask, d);
  if (rc != FIL_OK) return rc;
  if (N == 0) return FIL_OK;
  FIL_CHECK_ARG(q && k && v && dout && dq && dk && dv);
  const size_t sh1 = (size_t)(d.NA + d.NV) * d.KP * kPRS * sizeof(float);
  const size_t sh2 = (size_t)(d.NA + d.NV) * d.QP * kPRS * sizeof(float);
  if (pallow(pattn_q_kernel<1>, sh1) != FIL_OK || pallow(pattn_bwd_kv_kernel, sh2) != FIL_OK)
    return fail(FIL_ERR_UNSUPPORTED, "fil_pattn_bwd: Fq=%d Fk=%d need more than 160 KiB of LDS", Fq, Fk);
  ProfScope ps("pattn_bwd", (hipStream_t)stream, (double)N * 2.0 * Fq * (double)Fk * (4.0 * A + 4.0 * Av));
  hipLaunchKernelGGL(pattn_q_kernel<1>, dim3(N), dim3(256), sh1, (hipStream_t)stream, q, k, v, mask, dout, dq, d, scale);
  FIL_CHECK_LAUNCH();
  hipLaunchKernelGGL(pattn_bwd_kv_kernel, dim3(N), dim3(256), sh2, (hipStream_t)stream, q, k, v, mask, dout, dk, dv, d, scale);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
