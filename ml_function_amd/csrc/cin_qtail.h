// Quadratic tail of a three-layer CIN (L == 3, exact fp32): the small kernels around the pair-symmetric GEMMs.
//
// With p = L-2 and only sum-pools observable above layer p-1 (reference interactive_layer.py:322-325), the top two layers are
//   pool_L[m] = sum_h x1[m,h] R[m,h] + sum_f x[m,f] c[f] + sum_n bias_L[n],   R[m,h] = sum_{f',f} x[m,f'] x[m,f] T[(f',f),h]
//   pool_p[m] = sum_h x1[m,h] S[m,h] + sum_n bias_p[n],                       S[m,h] = sum_f x[m,f] wsum_p[(h,f)]
// with x1 = x^{p-1} (the first layer's feature map),  wsum_l[c] = sum_n W_l[c,n],
//   T[(f',f),h] = sum_n W_p[(h,f'),n] wsum_L[(n,f)],   c[f] = sum_n bias_p[n] wsum_L[(n,f)].
// R is a quadratic form in x for every h, i.e. a product over UNORDERED field pairs: the first layer's pair-symmetric GEMM
// kernels (forward, dW, dZ) run it unchanged with T in place of W_1 and x1 as the "gradient" operand -- F(F+1)/2 x H_{p-1}
// products per row instead of the H_{p-1} F (F+1) of the F+1-column form (cin_tail.h), and no column padding.  pool_p is the
// pooled-weights shortcut (cin_last_* kernels) applied to layer p.  This file: T / c, pool_L, and the chain from dT back to
// dW_p, dW_L and the biases.  Every sum runs in a fixed order.
#pragma once
#include "cin_kernels.h"

namespace fil {

constexpr int kQtConst = 64;   // cvec[f < F] = c[f], cvec[kQtConst] = sum_n bias_L[n]

// T[(f'*F + f)*Hpp + h] for block h < Hpp; block Hpp writes cvec and a zero bias vector for the R GEMM.
// LDS: wp [F][Hq+1] | wl [Hq][F]
static __global__ __launch_bounds__(256) void cin_qtail_t_kernel(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                                 const float* __restrict__ bias_p, const float* __restrict__ bias_L, int HL,
                                                                 float* __restrict__ T, float* __restrict__ cvec, float* __restrict__ zbias,
                                                                 int Hpp, int F, int Hq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int h = blockIdx.x;
  if (h == Hpp) {
    for (int f = threadIdx.x; f < F; f += 256) {
      float t = 0.f;
      for (int n = 0; n < Hq; ++n) t = fmaf(bias_p[n], wsumL[n * F + f], t);
      cvec[f] = t;
    }
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int n = 0; n < HL; ++n) t += bias_L[n];
      cvec[kQtConst] = t;
    }
    for (int i = threadIdx.x; i < Hpp; i += 256) zbias[i] = 0.f;
    return;
  }
  float* wp = smem;
  float* wl = smem + F * (Hq + 1);
  for (int i = threadIdx.x; i < F * Hq; i += 256) {
    const int fp = i / Hq, n = i - fp * Hq;
    wp[fp * (Hq + 1) + n] = Wp[((long)h * F + fp) * Hq + n];
  }
  for (int i = threadIdx.x; i < Hq * F; i += 256) wl[i] = wsumL[i];
  __syncthreads();
  for (int idx = threadIdx.x; idx < F * F; idx += 256) {
    const int fp = idx / F, f = idx - fp * F;
    const float* a = wp + fp * (Hq + 1);
    float t = 0.f;
    for (int n = 0; n < Hq; ++n) t = fmaf(a[n], wl[n * F + f], t);
    T[(long)idx * Hpp + h] = t;
  }
}

// pool_L[m] = sum_h xp[m,h] R[m,h] + sum_f x[m,f] cvec[f] + cvec[kQtConst]; 32 lanes per row.
static __global__ __launch_bounds__(256) void cin_qtail_pool_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                                    const float* __restrict__ R, int HS, const float* __restrict__ cvec,
                                                                    float* __restrict__ pool, int M, int F, int Hpp) {
  const int l = threadIdx.x & 31;
  const long row0 = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const float c0 = l < F ? cvec[l] : 0.f, c1 = l + 32 < F ? cvec[l + 32] : 0.f, cc = cvec[kQtConst];
  for (long m = row0; m < M; m += (long)gridDim.x * 8) {
    float t = 0.f;
    for (int h0 = 4 * l; h0 < Hpp; h0 += 128) {
      const float4 a = *reinterpret_cast<const float4*>(xpT + m * xps + h0);
      const float4 b = *reinterpret_cast<const float4*>(R + m * HS + h0);
      t = fmaf(a.x, b.x, t);
      if (h0 + 1 < Hpp) t = fmaf(a.y, b.y, t);
      if (h0 + 2 < Hpp) t = fmaf(a.z, b.z, t);
      if (h0 + 3 < Hpp) t = fmaf(a.w, b.w, t);
    }
    if (l < F) t = fmaf(xT[m * F + l], c0, t);
    if (l + 32 < F) t = fmaf(xT[m * F + l + 32], c1, t);
    t = half_wave_sum(t);
    if (l == 0) pool[m] = t + cc;
  }
}

// xs[m,f] = dP_L[m] x[m,f]  (the scaled factor of the pair products in the dT GEMM) and, per block of 256 rows, the column sums
// dcpart[blk][f] = sum_m xs[m,f] (-> dc[f] = d pool_L / d c[f]).  LDS: [256][F+1]
static __global__ __launch_bounds__(256) void cin_qtail_scale_kernel(const float* __restrict__ xT, const float* __restrict__ dP, int ldp, int K,
                                                                     float* __restrict__ xs, float* __restrict__ dcpart, int M, int F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const long r0 = (long)blockIdx.x * 256;
  const int nrow = (int)min((long)256, (long)M - r0);
  for (int i = threadIdx.x; i < 256 * F; i += 256) {
    const int rr = i / F, f = i - rr * F;
    float v = 0.f;
    if (rr < nrow) {
      const long m = r0 + rr, b = m / K;
      v = xT[m * F + f] * dP[b * ldp + (m - b * K)];
      xs[m * F + f] = v;
    }
    smem[rr * (F + 1) + f] = v;
  }
  __syncthreads();
  for (int f = threadIdx.x; f < F; f += 256) {
    float t = 0.f;
    for (int rr = 0; rr < 256; ++rr) t += smem[rr * (F + 1) + f];
    dcpart[(long)blockIdx.x * kQtConst + f] = t;
  }
}

// G[m,h] += dP_L[m] R[m,h]   (the pool_L part of the gradient of x^{p-1}); rows of both are 16-byte aligned
static __global__ __launch_bounds__(256) void cin_qtail_gadd_kernel(float* __restrict__ G, int HSg, const float* __restrict__ R, int HSr,
                                                                    const float* __restrict__ dP, int ldp, int K, int M, int Hpp) {
  const int q = (Hpp + 3) >> 2;
  const long total = (long)M * q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / q;
    const int h0 = 4 * (int)(i - m * q);
    const long b = m / K;
    const float d = dP[b * ldp + (m - b * K)];
    float4* g = reinterpret_cast<float4*>(G + m * HSg + h0);
    const float4 r = *reinterpret_cast<const float4*>(R + m * HSr + h0);
    float4 v = *g;
    v.x = fmaf(d, r.x, v.x);
    v.y = fmaf(d, r.y, v.y);
    v.z = fmaf(d, r.z, v.z);
    v.w = fmaf(d, r.w, v.w);
    *g = v;
  }
}

// dxT[m,f] += dP_L[m] (gxR[m,f] + dxR[m,f] + c[f]): the two halves of the quadratic form's gradient (pair-symmetric dZ kernel run on
// the UNSCALED x1) and the linear term
static __global__ __launch_bounds__(256) void cin_qtail_dx_kernel(float* __restrict__ dxT, const float* __restrict__ gxR, const float* __restrict__ dxR,
                                                                  const float* __restrict__ cvec, const float* __restrict__ dP, int ldp, int K,
                                                                  int M, int F) {
  const long total = (long)M * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int f = (int)(i - m * F);
    const long b = m / K;
    dxT[i] = fmaf(dP[b * ldp + (m - b * K)], (gxR[i] + dxR[i]) + cvec[f], dxT[i]);
  }
}

// Block h < Hpp:  dW_p[(h,f'),n] = v[(h,f')] + sum_f dT[(f',f),h] wsum_L[(n,f)]       (v: the pooled-weights shortcut's rank-one part,
//                 partL[h][(n,f)] = sum_f' W_p[(h,f'),n] dT[(f',f),h]                   given transposed, vT[f'][Hpp])
// LDS: wp [F][Hq+1] | wl [Hq][F] | dt [F][F+1]
static __global__ __launch_bounds__(256) void cin_qtail_params_kernel(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                                      const float* __restrict__ dT, const float* __restrict__ vT,
                                                                      float* __restrict__ dWp, float* __restrict__ partL, int Hpp, int F, int Hq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int h = blockIdx.x;
  float* wp = smem;
  float* wl = wp + F * (Hq + 1);
  float* dt = wl + Hq * F;
  for (int i = threadIdx.x; i < F * Hq; i += 256) {
    const int fp = i / Hq, n = i - fp * Hq;
    wp[fp * (Hq + 1) + n] = Wp[((long)h * F + fp) * Hq + n];
  }
  for (int i = threadIdx.x; i < Hq * F; i += 256) wl[i] = wsumL[i];
  for (int i = threadIdx.x; i < F * F; i += 256) {
    const int fp = i / F, f = i - fp * F;
    dt[fp * (F + 1) + f] = dT[(long)i * Hpp + h];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < F * Hq; idx += 256) {
    const int fp = idx / Hq, n = idx - fp * Hq;
    const float* d = dt + fp * (F + 1);
    const float* w = wl + n * F;
    float t = 0.f;
    for (int f = 0; f < F; ++f) t = fmaf(d[f], w[f], t);
    dWp[((long)h * F + fp) * Hq + n] = vT[(long)fp * Hpp + h] + t;
  }
  float* pl = partL + (long)h * Hq * F;
  for (int idx = threadIdx.x; idx < Hq * F; idx += 256) {
    const int n = idx / F, f = idx - n * F;
    float t = 0.f;
    for (int fp = 0; fp < F; ++fp) t = fmaf(wp[fp * (Hq + 1) + n], dt[fp * (F + 1) + f], t);
    pl[idx] = t;
  }
}

// dwsum_L[(n,f)] = sum_h partL[h][(n,f)] + bias_p[n] dc[f]  ->  dW_L[(n,f), n'] for every n' (64 rows (n,f) per workgroup);
// workgroup 0 also finishes dbias_p[n] = sum_m dP_p[m] + sum_f wsum_L[(n,f)] dc[f] and dbias_L[n'] = sum_m dP_L[m].
// dc[f] = sum of the ndc block partials of cin_qtail_scale_kernel; sp / sl: the nsl slice partials of dP_p / dP_L (cin_slice_sum_kernel).
static __global__ __launch_bounds__(256) void cin_qtail_fill_kernel(const float* __restrict__ partL, int Hpp, const float* __restrict__ dcpart, int ndc,
                                                                    const float* __restrict__ sp, const float* __restrict__ sl, int nsl,
                                                                    const float* __restrict__ bias_p, const float* __restrict__ wsumL,
                                                                    float* __restrict__ dWL, float* __restrict__ dbias_p, float* __restrict__ dbias_L,
                                                                    int F, int Hq, int HL) {
  __shared__ float dc[kQtConst];
  __shared__ float red[4][64];
  __shared__ float tot[2];
  __shared__ float val[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = threadIdx.x; f < F; f += 256) {
    float t = 0.f;
    for (int p = 0; p < ndc; ++p) t += dcpart[(long)p * kQtConst + f];
    dc[f] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 66) {
    const float* s = threadIdx.x == 64 ? sp : sl;
    float t = 0.f;
    for (int p = 0; p < nsl; ++p) t += s[p];
    tot[threadIdx.x - 64] = t;
  }
  const int C = Hq * F;
  const int c = blockIdx.x * 64 + lane;
  float t = 0.f;
  if (c < C)
    for (int h = wave; h < Hpp; h += 4) t += partL[(long)h * C + c];
  red[wave][lane] = t;
  __syncthreads();
  if (wave == 0 && c < C) {
    const int n = c / F, f = c - n * F;
    val[lane] = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + bias_p[n] * dc[f];
  }
  __syncthreads();
  // the 64 rows of this workgroup x HL columns, coalesced over the columns
  for (int i = threadIdx.x; i < 64 * HL; i += 256) {
    const int rr = i / HL, col = i - rr * HL;
    const int cr = blockIdx.x * 64 + rr;
    if (cr < C) dWL[(long)cr * HL + col] = val[rr];
  }
  if (blockIdx.x == 0) {
    for (int n = threadIdx.x; n < Hq; n += 256) {
      float u = tot[0];
      for (int f = 0; f < F; ++f) u = fmaf(wsumL[n * F + f], dc[f], u);
      dbias_p[n] = u;
    }
    for (int n = threadIdx.x; n < HL; n += 256) dbias_L[n] = tot[1];
  }
}

}  // namespace fil
