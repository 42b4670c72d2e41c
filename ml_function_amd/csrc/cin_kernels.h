// A3  xDeepFM CIN forward + backward for gfx950 (fp32 MFMA, v_mfma_f32_32x32x2_f32): device kernels.
//
// Replaces CIN.call of the reference (interactive_layer.py:310-327).  Per layer the reference materialises
// the outer product Z[b,k,c=h*F+f] = x^{l-1}[b,h,k] * x[b,f,k] (0.4-1.3 GB at the north-star shape),
// transposes it twice and runs a 1x1 Conv1D = GEMM [B*K, C] x [C, H].  Here Z never exists: every kernel is
// an implicit GEMM whose Z operand is regenerated in registers from the two small per-row vectors.
//
// GEMM view: rows m = b*K + k (M = B*K), reduction c = (h,f) (C = Hp*F), columns n (H).
//   fwd   cin_fwd3_kernel   X^l[m,n]  = sum_c Z[m,c] W[c,n] + bias[n]         (A = Z generated, B = W streamed)
//   bwd   cin_dw3_kernel    dW[c,n]   = sum_m Z[m,c] G[m,n]                   (A = Z^T generated, B = G streamed)
//   bwd   cin_dz3_kernel    dZ[c,m]   = sum_n W[c,n] G[m,n], consumed in registers:
//                           G^{l-1}[m,h] = sum_f dZ[(h,f),m] x[m,f];  dX[m,f] += sum_h dZ[(h,f),m] x^{l-1}[m,h]
// The reduction order of a GEMM is free, so each kernel picks the order that makes its generated operand
// lane-local.  All three are MFMA-bound: fp32 MFMA issues one 32x32x2 tile per 64 cycles per SIMD.
//
// MFMA 32x32x2 f32 operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator register r of lane l holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
#pragma once
#include "common.h"

namespace fil {

constexpr int kCinThreads = 256;  // 4 waves per workgroup, one per SIMD
constexpr int kCinMaxL = 8;
constexpr int kCinMaxH = 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Makes the compiler wait HERE for any pending load of v (an empty asm that "uses" the register), so that no
// s_waitcnt vmcnt(0) is placed later inside an MFMA loop, where it would also drain the prefetch just issued.
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }
typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void settle(f32x4s& v) { asm volatile("" : "+v"(v)); }

// Raw buffer descriptor over [p, p + bytes): loads take a constant per-lane 32-bit voffset plus a scalar soffset, and anything
// past the end reads as zero through the range check -- no per-load 64-bit address arithmetic, no clamping.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)std::min<long>(bytes, 0x7fffffffL), 0x00020000);
}
// the same from values the compiler cannot prove wave-uniform (derived from threadIdx >> 6): pinned to scalar registers, otherwise
// every load through the descriptor is wrapped in a waterfall loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const float* p, long bytes) {
  const unsigned long a = reinterpret_cast<unsigned long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)std::min<long>(bytes, 0x7fffffffL));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long)hi << 32) | lo), 0, n, 0x00020000);
}

// #################################################################################################
// v3 kernels: m-major internal layouts + LDS-free streaming GEMMs.
//
// Internal tensors are "m-major": row m = b*K + k.   xT [M][F];  feature maps / their gradients [M][HS] with
// HS = 128*ceil(H/128) (columns >= H are zero).  These are the natural operand/result layouts of the MFMA tiles:
// a lane that owns column j of a 32x32 tile is given the 4 consecutive feature maps n = 4j..4j+3 of a 128-column
// chunk (one per accumulator block), so every B operand fetch / result store is one aligned 16-byte access.
// W is re-packed per call into zero-padded, step-ordered copies (2.6 MB, a few microseconds) so that every wave
// streams its operands from L2 with a register prefetch queue: no LDS staging, no workgroup barriers.
// The x fragment of a row lives in registers, which needs a compile-time step count per h: JT = ceil(F/2) rounded
// up to a multiple of 4 (menu 4..32; the padding steps multiply by zero).
// #################################################################################################
constexpr int kQDepthMax = 10;

// x [B,F,K] -> xT [M][F]   (one workgroup per sample, transposed through LDS)
// x2T != nullptr: also the wrapped rows of x, x2[m][p] = x[m][p mod F] for p < XL = F + 2*JTs, POSITION-major in blocks of 64 rows:
//   x2T[((m >> 6) * XL + p) * 64 + (m & 63)]
// The pair-symmetric forward kernel reads x[m, (h + d) mod F] as x2[m][h + d]: a per-lane constant offset plus a SCALAR offset h (no
// per-load index arithmetic), and the 32 rows of a wave half are 128 contiguous bytes (read row-major, [M][F], every such load
// touched 64 different lines: at 22 loads per h the vector memory pipeline, not the MFMA, set that kernel's pace -- 0.128 ms against
// 0.098 with the loads removed).
__device__ __forceinline__ long cin_x2_index(long m, int p, int XL) { return ((m >> 6) * XL + p) * 64 + (m & 63); }
inline __host__ __device__ size_t cin_x2_floats(long M, int XL) { return (size_t)((M + 63) >> 6) * XL * 64; }
__device__ __forceinline__ void cin_transpose_in_body(const float* __restrict__ x, float* __restrict__ xT, int F, int K, long b, float* smem,
                                                      float* __restrict__ x2T = nullptr, int XL = 0) {
  const float* src = x + b * F * K;
  for (int i = threadIdx.x; i < F * K; i += 256) smem[(i / K) * (K + 1) + (i % K)] = src[i];
  __syncthreads();
  float* dst = xT + b * K * F;
  for (int i = threadIdx.x; i < F * K; i += 256) dst[i] = smem[(i % F) * (K + 1) + (i / F)];
  if (x2T != nullptr) {
    for (int i = threadIdx.x; i < XL * K; i += 256) {
      const int p = i / K, k = i - p * K;
      x2T[cin_x2_index(b * K + k, p, XL)] = smem[(p % F) * (K + 1) + k];
    }
  }
}
// The same for one 64-row block of xT / x2T per workgroup when K divides 64 (a block is 64 / K whole samples): the block's 64 F inputs,
// its 64 F floats of xT and its 64 XL floats of x2T are each ONE contiguous range -- 16-byte accesses, every line written whole by
// one workgroup (per sample: 4-byte accesses, and a 256-byte line of x2T assembled from four workgroups' quarter writes).
// ks = log2 K.  LDS: [64 / K][F][K + 1].  M = B K need not be a multiple of 64 (the last block's missing samples are skipped).
__device__ __forceinline__ void cin_transpose_block_body(const float* __restrict__ x, float* __restrict__ xT, float* __restrict__ x2T, int F, int ks,
                                                         long blk, long M, int XL, float* smem) {
  const int K = 1 << ks, kp = K + 1;
  const long m0 = blk * 64;
  const int rows = (int)min((long)64, M - m0);   // a multiple of K
  const int n = rows * F;                        // floats of the block in x and in xT (a multiple of 4 only if rows F is: scalar tail)
  const float* src = x + m0 * F;                 // (m0 = b0 K: sample b0 starts at x + b0 F K)
  const int n4 = n >> 2;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i0 + u * 256 < n4 ? reinterpret_cast<const float4*>(src)[i0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * (i0 + u * 256);
      if (i < 4 * n4) {
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) smem[((i + c) >> ks) * kp + ((i + c) & (K - 1))] = e[c];   // x index (s F + f) K + k -> (s F + f)(K + 1) + k
      }
    }
  }
  for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) smem[(i >> ks) * kp + (i & (K - 1))] = src[i];
  __syncthreads();
  // xT: floats m0 F + i, i = j F + f  <-  image[(s F + f)(K + 1) + k], j = s K + k
  float* dst = xT + m0 * F;
  for (int i4 = threadIdx.x; i4 < n4; i4 += 256) {
    int j = (4 * i4) / F, f = 4 * i4 - j * F;
    float e[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[c] = smem[((j >> ks) * F + f) * kp + (j & (K - 1))];
      if (++f == F) {
        f = 0;
        ++j;
      }
    }
    reinterpret_cast<float4*>(dst)[i4] = make_float4(e[0], e[1], e[2], e[3]);
  }
  for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) {
    const int j = i / F, f = i - j * F;
    dst[i] = smem[((j >> ks) * F + f) * kp + (j & (K - 1))];
  }
  if (x2T != nullptr) {
    // x2T: floats (blk XL + p) 64 + j  <-  x[m0 + j][p mod F]; four consecutive rows j per thread (K >= 4: one sample, consecutive k)
    float* d2 = x2T + blk * XL * 64;
    for (int i4 = threadIdx.x; i4 < XL * 16; i4 += 256) {
      const int p = i4 >> 4, j0 = (i4 & 15) * 4;
      int f = p;
      while (f >= F) f -= F;
      float e[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        e[c] = j < rows ? smem[((j >> ks) * F + f) * kp + (j & (K - 1))] : 0.f;
      }
      if (j0 + 3 < rows) reinterpret_cast<float4*>(d2)[i4] = make_float4(e[0], e[1], e[2], e[3]);
      else
        for (int c = 0; c < 4; ++c)
          if (j0 + c < rows) d2[4 * i4 + c] = e[c];
    }
  }
}
// The wrapped rows of one 64-row block from an input that is already transposed (FIL_CIN_X_TRANSPOSED; any K): the block's 64 F
// floats of xT are one contiguous range, its 64 XL floats of x2T another -- through LDS [64][F | 1] (odd row pitch: the column reads
// are conflict-free), 16-byte stores.
__device__ __forceinline__ void cin_wrap_block_body(const float* __restrict__ xT, float* __restrict__ x2T, int F, long blk, long M, int XL,
                                                    float* smem) {
  const int ld = F | 1;
  const long m0 = blk * 64;
  const int rows = (int)min((long)64, M - m0);
  const float* src = xT + m0 * F;
  const float rF = 1.f / (float)F;
  for (int i = threadIdx.x; i < rows * F; i += 256) {   // (coalesced dwords: xT as given need not be 16-byte aligned)
    const int j = (int)(((float)i + 0.5f) * rF);
    smem[j * ld + (i - j * F)] = src[i];
  }
  __syncthreads();
  float* d2 = x2T + blk * XL * 64;
  for (int i4 = threadIdx.x; i4 < XL * 16; i4 += 256) {
    const int p = i4 >> 4, j0 = (i4 & 15) * 4;
    int f = p;
    while (f >= F) f -= F;
    float e[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) e[c] = j0 + c < rows ? smem[(j0 + c) * ld + f] : 0.f;
    if (j0 + 3 < rows) reinterpret_cast<float4*>(d2)[i4] = make_float4(e[0], e[1], e[2], e[3]);
    else
      for (int c = 0; c < 4; ++c)
        if (j0 + c < rows) d2[4 * i4 + c] = e[c];
  }
}
// the same from an input that is already transposed (FIL_CIN_X_TRANSPOSED): the K rows of sample b
__device__ __forceinline__ void cin_wrap_rows_body(const float* __restrict__ xT, float* __restrict__ x2T, int F, int K, long b, int XL) {
  const float* src = xT + b * K * F;
  for (int i = threadIdx.x; i < XL * K; i += 256) {
    const int p = i / K, k = i - p * K;
    x2T[cin_x2_index(b * K + k, p, XL)] = src[k * F + p % F];
  }
}
static __global__ __launch_bounds__(256) void cin_transpose_in_kernel(const float* __restrict__ x, float* __restrict__ xT, int F, int K,
                                                               float* __restrict__ x2T = nullptr, int XL = 0, int xt_in = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [F][K+1]
  if (xt_in) cin_wrap_rows_body(x, x2T, F, K, blockIdx.x, XL);
  else cin_transpose_in_body(x, xT, F, K, blockIdx.x, smem, x2T, XL);
}

// dx [B,F,K] = dxT [M][F] (+ addT [M][F])
// qc != nullptr (quadratic tail): + scale[m] * (q1[m,f] + q2[m,f] + qc[f]), scale[m] = qs[b*ldq + k] -- the two halves of the
// quadratic form's gradient, computed on the unscaled x1 (q1 == nullptr: the merged form has them in dxT already), and its linear term
static __global__ __launch_bounds__(256) void cin_transpose_out_kernel(const float* __restrict__ dxT, const float* __restrict__ addT,
                                                                float* __restrict__ dx, int F, int K, const float* __restrict__ q1 = nullptr,
                                                                const float* __restrict__ q2 = nullptr, const float* __restrict__ qc = nullptr,
                                                                const float* __restrict__ qs = nullptr, int ldq = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][F+1]
  const long b = blockIdx.x;
  const float* src = dxT + b * K * F;
  const float* src2 = addT != nullptr ? addT + b * K * F : nullptr;
  if (qc != nullptr) {
    const float* a1 = q1 != nullptr ? q1 + b * K * F : nullptr;
    const float* a2 = q1 != nullptr ? q2 + b * K * F : nullptr;
    for (int i = threadIdx.x; i < F * K; i += 256) {
      const int k = i / F, f = i - k * F;
      smem[k * (F + 1) + f] = (src[i] + (src2 ? src2[i] : 0.f)) + qs[b * ldq + k] * ((a1 ? a1[i] + a2[i] : 0.f) + qc[f]);
    }
  } else
  for (int i = threadIdx.x; i < F * K; i += 256) smem[(i / F) * (F + 1) + (i % F)] = src[i] + (src2 ? src2[i] : 0.f);
  __syncthreads();
  float* dst = dx + b * F * K;
  for (int i = threadIdx.x; i < F * K; i += 256) dst[i] = smem[(i % K) * (F + 1) + (i / K)];
}

// Wf[chunk][h][fp < 2*JT][128] = W[(h*F + fp)*H + chunk*128 + col]  (zero rows fp >= F, zero columns n >= H)
static __global__ __launch_bounds__(256) void cin_pack_wf_kernel(const float* __restrict__ W, float* __restrict__ Wf, int Hp, int F, int H,
                                                          int JT2, int chunks) {
  const long total = (long)chunks * Hp * JT2 * 128;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i & 127);
    long t = i >> 7;
    const int fp = (int)(t % JT2);
    t /= JT2;
    const int h = (int)(t % Hp), chunk = (int)(t / Hp);
    const int n = chunk * 128 + col;
    Wf[i] = (fp < F && n < H) ? W[((long)h * F + fp) * H + n] : 0.f;
  }
}

// dW[(h,f),n] = dW[(f,h),n] = Dsym[pair(h,f)][n]: expands the pair-indexed first-layer weight gradient
static __global__ __launch_bounds__(256) void cin_expand_sym_kernel(const float* __restrict__ dsym, float* __restrict__ dW, int F, int D, int H) {
  const long total = (long)F * F * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % H);
    const int c = (int)(i / H);
    const int h = c / F, f = c - h * F;
    const int d1 = f >= h ? f - h : f - h + F;
    const long pair = d1 <= F / 2 ? (long)h * D + d1 : (long)f * D + (F - d1);
    dW[i] = dsym[pair * H + n];
  }
}

// Symmetric first layer: Wf[chunk][h][d < JT2][128] = pair weight of (h, f = (h+d) mod F): W[(h,h)] for d = 0,
// W[(h,f)] + W[(f,h)] for 0 < d <= F/2 (halved at d = F/2 when F is even: that pair is met from both ends), else 0.
__device__ __forceinline__ void cin_pack_wf_sym_body(const float* __restrict__ W, float* __restrict__ Wf, int F, int H, int JT2, int chunks,
                                                     int bid, int nblocks) {
  const long total = (long)chunks * F * JT2 * 128;
  for (long i = (long)bid * 256 + threadIdx.x; i < total; i += (long)nblocks * 256) {
    const int col = (int)(i & 127);
    long t = i >> 7;
    const int d = (int)(t % JT2);
    t /= JT2;
    const int h = (int)(t % F), chunk = (int)(t / F);
    const int n = chunk * 128 + col;
    float v = 0.f;
    if (n < H && d <= F / 2) {
      const int f = (h + d) % F;
      if (d == 0) v = W[((long)h * F + h) * H + n];
      else {
        v = W[((long)h * F + f) * H + n] + W[((long)f * F + h) * H + n];
        if (2 * d == F) v *= 0.5f;
      }
    }
    Wf[i] = v;
  }
}
static __global__ __launch_bounds__(256) void cin_pack_wf_sym_kernel(const float* __restrict__ W, float* __restrict__ Wf, int F, int H, int JT2,
                                                              int chunks) {
  cin_pack_wf_sym_body(W, Wf, F, H, JT2, chunks, blockIdx.x, gridDim.x);
}

// Forward layer, streaming form.  Wave = 32*MB rows x 128 columns (one chunk); step (h, j): half 0 / 1 take
// f = 2j / 2j+1; the W row pair of step s = h*JT + j is Wf row 2s+half, so the B-operand stream is linear.
// x^{l-1}[m,h] is one dword per h (prefetched); the queue holds DEPTH steps of B operands (16 B per lane each).
//
// SYM (first layer only, x^{l-1} = x): Z[m,(h,f)] = x[m,h] x[m,f] is symmetric in (h,f), so the reduction runs over the
// F*(F/2+1) unordered pairs (h, f = (h+d) mod F), d = 0..F/2, against pre-summed weights W[(h,f)] + W[(f,h)]
// (cin_pack_wf_sym_kernel): half the MFMA work.  Step (h, j) then multiplies by x[m,(h + 2j + half) mod F], which
// moves with h, so the x fragment is re-fetched per h (one h ahead) instead of living in registers for the whole run.
//
// KS > 1 (small M: fewer row blocks than SIMDs): KS waves of the workgroup share one row block and split the
// reduction over h between them; the partial accumulators are folded through LDS in wave order and the group's first wave runs
// the epilogue (see cin_tail_fwd_kernel).
template <int MB, int JT, bool SYM = false, int KS = 1>
__global__ __launch_bounds__(256, 1) void cin_fwd3_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                          const float* __restrict__ Wf, const float* __restrict__ bias,
                                                          float* __restrict__ xoutT, int HS, float* __restrict__ pool_part, int M,
                                                          int F, int Hp, int H, const float* __restrict__ wsn,
                                                          const float* __restrict__ bias_next, int H_next,
                                                          float* __restrict__ pool_next) {
  // queue depth: the largest divisor of JT not above kQDepthMax (slot j % DEPTH must mean the same step in every h)
  constexpr int DEPTH = JT <= kQDepthMax ? JT : (JT % 10 == 0 ? 10 : (JT % 8 == 0 ? 8 : (JT % 7 == 0 ? 7 : (JT % 6 == 0 ? 6 : 4))));
  static_assert(JT % DEPTH == 0, "queue depth must divide the steps per h");
  const int tid = threadIdx.x, lane = tid & 63, wave = KS == 1 ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int chunk = blockIdx.y;
  const int kpart = wave % KS;
  const int wrow0 = ((blockIdx.x * 4 + wave) / KS) * (32 * MB);
  if (KS == 1 && wrow0 >= M) return;  // whole wave past the end (no barriers in this kernel unless KS > 1)
  const int hq_ = (Hp + KS - 1) / KS;
  const int h_lo = KS == 1 ? 0 : min(Hp, kpart * hq_);
  const int h_hi = KS == 1 ? Hp : (wrow0 < M ? min(Hp, h_lo + hq_) : h_lo);
  const int hs = min(h_lo, Hp - 1);
  long mq[MB];
  bool vq[MB];
  float xr[MB][JT];
  float xn[SYM ? MB : 1][SYM ? JT : 1];
  int d0[SYM ? JT : 1];  // (2j + half) mod F
  if constexpr (SYM) {
#pragma unroll
    for (int j = 0; j < JT; ++j) d0[j] = (2 * j + half) % F;
  }
  // SYM: x[m, (h + 2j + half) mod F] for every j (rows past M are clamped: computed, never stored)
  auto load_x = [&](int h, float (&dst)[SYM ? MB : 1][SYM ? JT : 1]) {
    if constexpr (SYM) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float* xrow = xT + mq[mb] * F;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          int idx = h + d0[j];
          idx -= idx >= F ? F : 0;
          dst[mb][j] = xrow[idx];
        }
      }
    }
  };
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wrow0 + mb * 32 + r;
    vq[mb] = m < M;
    mq[mb] = vq[mb] ? m : M - 1;
    if constexpr (!SYM) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int f = 2 * j + half;
        xr[mb][j] = (vq[mb] && f < F) ? xT[mq[mb] * F + f] : 0.f;
      }
    }
  }
  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  if constexpr (SYM) {
    // Exact pair-symmetric first layer.  Step (h, j) of lane half `half` multiplies x[m,h] by x[m,(h + 2j + half) mod F], read from the
    // WRAPPED, position-major rows x2 (cin_transpose_in_body; xpT / xps carry x2T / XL here) as x2[m][h + 2j + half]: the lane part of
    // the address is a constant, h a scalar offset and 2j an immediate -- raw buffer loads, no per-load index arithmetic, two lines per
    // load instruction.  The W stream is linear in the step s = h*JT + j: one descriptor, scalar offsets.
    // Two values of h per loop iteration: the fragment of h+1 lands in the registers h-1 released (no copies).
    const int XL = xps;
    const int wrow_u = __builtin_amdgcn_readfirstlane(wrow0);
    // this wave's 64-row block of the wrapped rows ([p][64 rows]; a 32-row wave takes one half of it)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(xpT + (long)(wrow_u >> 6) * XL * 64, (long)XL * 256);
    const int chunk_u = __builtin_amdgcn_readfirstlane(chunk);
    const long wbytes = (long)Hp * (2 * JT) * 128 * 4;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc_uniform(Wf + (long)chunk_u * (wbytes >> 2), wbytes);     // steps past the end read zeros
    const int wo = (half * 32 + r) * 16;
    int vrow[MB], vhalf[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      vrow[mb] = ((wrow_u & 63) + mb * 32 + r) * 4;
      vhalf[mb] = vrow[mb] + half * 256;
    }
    auto ldw = [&](int s) {   // B operands of step s (uniform)
      return __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rw, wo, s * 1024, 0));
    };
    auto ldfrag = [&](int h, float (&xf)[MB][JT], float (&xp)[MB]) {
      const int hb = __builtin_amdgcn_readfirstlane(h) * 256;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        xp[mb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vrow[mb], hb, 0));
#pragma unroll
        for (int j = 0; j < JT; ++j) xf[mb][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vhalf[mb] + 512 * j, hb, 0));
      }
    };
    f32x4s q[DEPTH];
    const int s_lo = hs * JT;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = ldw(s_lo + d);
    float xa[MB][JT], xb[MB][JT], pa[MB], pb[MB];
    ldfrag(hs, xa, pa);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) settle(q[d]);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) settle(pa[mb]);
    float ac[MB], an[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) ac[mb] = pa[mb] * xa[mb][0];
    // one value of h: fragment (xc, pc) in use, (xn_, pn) being fetched for h + 1
    auto run_h = [&](int h, float (&xc)[MB][JT], float (&pc)[MB], float (&xn_)[MB][JT], float (&pn)[MB]) {
      ldfrag(min(h + 1, Hp - 1), xn_, pn);
      const int sb = h * JT;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const f32x4s w = q[j % DEPTH];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) an[mb] = j + 1 < JT ? pc[mb] * xc[mb][j + 1 < JT ? j + 1 : 0] : pn[mb] * xn_[mb][0];
        __builtin_amdgcn_sched_barrier(0);   // (these three pin the step: without them the kernel runs 0.135 ms instead of 0.103)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float a = ac[mb];
          acc[mb][0] = mfma32(a, w[0], acc[mb][0]);
          acc[mb][1] = mfma32(a, w[1], acc[mb][1]);
          acc[mb][2] = mfma32(a, w[2], acc[mb][2]);
          acc[mb][3] = mfma32(a, w[3], acc[mb][3]);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) ac[mb] = an[mb];
        __builtin_amdgcn_sched_barrier(0);
        q[j % DEPTH] = ldw(sb + j + DEPTH);   // (after the step's MFMAs: it may land in the registers it replaces)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int h = h_lo;
#pragma unroll 1
    for (; h + 1 < h_hi; h += 2) {
      run_h(h, xa, pa, xb, pb);
      run_h(h + 1, xb, pb, xa, pa);
    }
    if (h < h_hi) run_h(h, xa, pa, xb, pb);
  } else {
  const float4* wbase = reinterpret_cast<const float4*>(Wf + (long)chunk * Hp * (2 * JT) * 128) + (half * 32 + r);
    float4 q[DEPTH];
  #pragma unroll
    for (int d = 0; d < DEPTH; ++d) q[d] = wbase[(long)hs * (2 * JT) * 32 + (long)(2 * d) * 32];
    const float* xprow[MB];
    float xpv[MB], xpn[MB];
  #pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      xprow[mb] = xpT + mq[mb] * xps;
      xpv[mb] = xprow[mb][hs];
      xpn[mb] = 0.f;
    }
    // Every prologue load is waited for HERE, once.  The compiler orders the prologue's loads as it likes; when the operand of
    // the loop's first step is not the oldest of them, the wait count it derives for the top of the loop (the merge of "from
    // the prologue" and "from the previous iteration") collapses to a few, i.e. it drains the whole operand queue at the top
    // of every h: ~1,500 idle cycles per 10,240 cycles of MFMA work.  With nothing pending on entry the count at the loop top
    // comes from the back edge alone (the 9 younger refills + this iteration's loads stay in flight).  A/B on one box, c4 l2:
    // 0.659 -> 0.634 ms.
  #pragma unroll
    for (int d = 0; d < DEPTH; ++d) settle(q[d].x);
  #pragma unroll
    for (int mb = 0; mb < MB; ++mb) settle(xpv[mb]);
    float ac[MB], an[MB];
  #pragma unroll
    for (int mb = 0; mb < MB; ++mb) ac[mb] = xpv[mb] * xr[mb][0];
  #pragma unroll 1
    for (int h = h_lo; h < h_hi; ++h) {
      const bool more = h + 1 < Hp;
      // (branch-free: a conditional load makes the compiler's wait-count bookkeeping fall back to vmcnt(0) at the top of
      // every h, which drains the whole operand queue once per h -- ~1,500 idle cycles against 10,240 of MFMA work)
  #pragma unroll
      for (int mb = 0; mb < MB; ++mb) xpn[mb] = xprow[mb][more ? h + 1 : h];
      if constexpr (SYM) load_x(more ? h + 1 : h, xn);
      const float4* wh = wbase + (long)h * (2 * JT) * 32;
      const float4* whn = wbase + (long)(more ? h + 1 : h) * (2 * JT) * 32;
  #pragma unroll
      for (int j = 0; j < JT; ++j) {
        const float4 w = q[j % DEPTH];
        // the A operands of step j+1 as one block in front of step j's MFMAs: a v_mul whose result feeds the very next MFMA costs
        // its VALU -> MFMA operand wait states on the matrix pipe (tools/probe_mfma16.hip)
  #pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (j + 1 < JT) an[mb] = xpv[mb] * xr[mb][j + 1];
          else an[mb] = xpn[mb] * (SYM ? xn[SYM ? mb : 0][0] : xr[mb][0]);
        }
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float a = ac[mb];
          acc[mb][0] = mfma32(a, w.x, acc[mb][0]);
          acc[mb][1] = mfma32(a, w.y, acc[mb][1]);
          acc[mb][2] = mfma32(a, w.z, acc[mb][2]);
          acc[mb][3] = mfma32(a, w.w, acc[mb][3]);
        }
  #pragma unroll
        for (int mb = 0; mb < MB; ++mb) ac[mb] = an[mb];
        __builtin_amdgcn_sched_barrier(0);
        // refill this slot with the operand of step j + DEPTH (possibly in the next h) -- AFTER the step's MFMAs were issued,
        // so the load may land in the registers it replaces: when JT == DEPTH every slot is refilled once per h and a refill
        // into fresh registers has to be copied back at the end of the h iteration, which waits for ALL of them (vmcnt(0))
        const int jn = j + DEPTH;
        q[j % DEPTH] = jn < JT ? wh[(long)(2 * jn) * 32] : whn[(long)(2 * (jn - JT)) * 32];
        __builtin_amdgcn_sched_barrier(0);  // keep the refill load here (the scheduler would sink it to its use)
      }
  #pragma unroll
      for (int mb = 0; mb < MB; ++mb) xpv[mb] = xpn[mb];
      if constexpr (SYM) {
  #pragma unroll
        for (int mb = 0; mb < MB; ++mb)
  #pragma unroll
          for (int j = 0; j < JT; ++j) xr[mb][j] = xn[mb][j];
      }
    }
  }

  if constexpr (KS > 1) {
    __shared__ float fold[3][MB * 64][64];   // (wave 0 never parks)
    if (kpart > 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) fold[wave - 1][(mb * 4 + nb) * 16 + e][lane] = acc[mb][nb][e];
    }
    __syncthreads();
    if (kpart > 0 || wrow0 >= M) return;
#pragma unroll
    for (int k = 1; k < KS; ++k)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mb][nb][e] += fold[wave + k - 1][(mb * 4 + nb) * 16 + e][lane];
  }
  // ---- epilogue: + bias, store [M][HS] (lane r owns columns 4r..4r+3 of the chunk), sum-pool over columns
  float bv[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = chunk * 128 + 4 * r + nb;
    bv[nb] = n < H ? bias[n] : 0.f;
  }
  const float bsum = half_wave_sum_hi(bv[0] + bv[1] + bv[2] + bv[3]);  // (valid on the writer lane r == 31)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = wrow0 + mb * 32 + mfma32_row(reg, half);
      const float v0 = acc[mb][0][reg], v1 = acc[mb][1][reg], v2 = acc[mb][2][reg], v3 = acc[mb][3][reg];
      if (xoutT != nullptr && m < M)
        *reinterpret_cast<float4*>(xoutT + (long)m * HS + chunk * 128 + 4 * r) = make_float4(v0 + bv[0], v1 + bv[1], v2 + bv[2], v3 + bv[3]);
      const float p = half_wave_sum_hi((v0 + v1) + (v2 + v3)) + bsum;
      if (r == 31 && m < M) pool_part[(long)chunk * M + m] = p;
    }
  }

  // ---- fused sum-pool of the NEXT layer when that is the last one (mode 0; see the cin_last_* kernels):
  //   p_next[m] = sum_n x^l[m,n] * t[m,n],  t[m,n] = sum_f x[m,f] wsum_next[n,f]
  // t is one more small MFMA product from the x fragment already in registers; wsn is packed like Wf ([2*JT][128]
  // per chunk), so t's accumulators line up element for element with this layer's output tile.
  if constexpr (!SYM) {
    if (wsn != nullptr) {
      float bn = 0.f;
      if (chunk == 0) {
        for (int n = lane; n < H_next; n += 64) bn += bias_next[n];
        bn = wave_sum(bn);
      }
      const float4* wsb = reinterpret_cast<const float4*>(wsn + (long)chunk * (2 * JT) * 128) + (half * 32 + r);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        f32x16 t[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) t[nb][i] = 0.f;
        // (4-deep operand queue: an unprefetched 16-byte load per step would pay the L2 latency JT times)
        constexpr int QD = JT < 4 ? JT : 4;
        float4 wq[QD];
#pragma unroll
        for (int j = 0; j < QD; ++j) wq[j] = wsb[(long)(2 * j) * 32];
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const float4 w = wq[j % QD];
          if (j + QD < JT) wq[j % QD] = wsb[(long)(2 * (j + QD)) * 32];
          const float a = xr[mb][j];
          t[0] = mfma32(a, w.x, t[0]);
          t[1] = mfma32(a, w.y, t[1]);
          t[2] = mfma32(a, w.z, t[2]);
          t[3] = mfma32(a, w.w, t[3]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = wrow0 + mb * 32 + mfma32_row(reg, half);
          float e = (acc[mb][0][reg] + bv[0]) * t[0][reg];
          e = fmaf(acc[mb][1][reg] + bv[1], t[1][reg], e);
          e = fmaf(acc[mb][2][reg] + bv[2], t[2][reg], e);
          e = fmaf(acc[mb][3][reg] + bv[3], t[3][reg], e);
          const float p = half_wave_sum_hi(e) + bn;
          if (r == 31 && m < M) pool_next[(long)chunk * M + m] = p;
        }
      }
    }
  }
}

// Wz[(t*32 + i)*NCOL + col]: W rows in the "slot" order of the dZ kernel.  Tile row i <-> (slot 16t + rr, parity hf),
// rr = (i&3) + 4*(i>>3), hf = (i>>2)&1;  slot -> (h, j) = (slot / JT, slot % JT), f = 2j + hf;  zero when out of range.
static __global__ __launch_bounds__(256) void cin_pack_wz_kernel(const float* __restrict__ W, float* __restrict__ Wz, int Hp, int F, int H,
                                                          int JT, int NCOL, int tiles) {
  const long total = (long)tiles * 32 * NCOL;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx % NCOL);
    const long row = idx / NCOL;
    const int i = (int)(row & 31);
    const long t = row >> 5;
    const int rr = (i & 3) + 4 * (i >> 3), hf = (i >> 2) & 1;
    const long slot = 16 * t + rr;
    const int h = (int)(slot / JT), j = (int)(slot - (long)h * JT);
    const int f = 2 * j + hf;
    Wz[idx] = (h < Hp && f < F && col < H) ? W[((long)h * F + f) * H + col] : 0.f;
  }
}

// Symmetric first layer: slot (h, j), parity hf <-> pair (h, f = (h + d) mod F), d = 2j + hf; weights as in
// cin_pack_wf_sym_kernel.
__device__ __forceinline__ void cin_pack_wz_sym_body(const float* __restrict__ W, float* __restrict__ Wz, int F, int H, int JT, int NCOL,
                                                     int tiles, int bid, int nblocks) {
  const long total = (long)tiles * 32 * NCOL;
  for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblocks * 256) {
    const int col = (int)(idx % NCOL);
    const long row = idx / NCOL;
    const int i = (int)(row & 31);
    const long t = row >> 5;
    const int rr = (i & 3) + 4 * (i >> 3), hf = (i >> 2) & 1;
    const long slot = 16 * t + rr;
    const int h = (int)(slot / JT), j = (int)(slot - (long)h * JT);
    const int d = 2 * j + hf;
    float v = 0.f;
    if (h < F && d <= F / 2 && col < H) {
      const int f = (h + d) % F;
      if (d == 0) v = W[((long)h * F + h) * H + col];
      else {
        v = W[((long)h * F + f) * H + col] + W[((long)f * F + h) * H + col];
        if (2 * d == F) v *= 0.5f;
      }
    }
    Wz[idx] = v;
  }
}
static __global__ __launch_bounds__(256) void cin_pack_wz_sym_kernel(const float* __restrict__ W, float* __restrict__ Wz, int F, int H, int JT,
                                                              int NCOL, int tiles) {
  cin_pack_wz_sym_body(W, Wz, F, H, JT, NCOL, tiles, blockIdx.x, gridDim.x);
}

constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// Backward data path, streaming form.  Wave = 32*MB rows m (on the lanes).  dZ^T tile = Wz tile (32 slot rows,
// A operand streamed from L2: 16 bytes per 4 steps) x G^T (B operand: the lane's G row, NHMAX registers per 32 rows).
// Slot order (see cin_pack_wz_kernel) makes accumulator register rr of tile t the channel (h, f = 2j+half) with
// 16t+rr = h*JT + j, so with a compile-time JT the whole contraction is register-only:
//   gx       += dZ * x[m,f]            -> G^{l-1}[m,h] when the h is complete (halves added with one shuffle)
//   dxacc[j] += dZ * x^{l-1}[m,h]      -> dX[m, 2j+half]
// The (tile, register) -> (h, j) pattern repeats every P = JT/gcd(16,JT) tiles = HPP = 16P/JT values of h.
//
// SYM (first layer, x^{l-1} = x, weights pre-summed over (h,f)/(f,h) by cin_pack_wz_sym_kernel): slot (h, j) is the
// unordered pair (h, f = (h + 2j + half) mod F).  f now moves with h, so the x fragment / dX accumulators are
// indexed at run time: they live in LDS as [mb][f][row] (FR rows of kSymStride floats per mb, one column per row m,
// shared by the two lane halves of a row -- within one instruction the halves touch f and f+1, never the same word).
//   gx      += dZ * x[m,f]  -> Gx[m,h]  (gx0T)        dxs[f] += dZ * x[m,h]  -> dX[m,f]   (summed by transpose_out)
constexpr int kSymStride = 160;  // 4 waves x 32 rows + 32: consecutive f land in opposite bank halves
// G^{l-1} leaves through a per-wave LDS line buffer [32 rows][kGlStride]: a lane completes HPP consecutive columns of its
// row at a time, and storing those directly is a 4-16 byte write at a 512-byte row stride -- every 128-byte line of G^{l-1}
// is then written in 8-32 pieces spread over the whole kernel, L2 evicts it half-filled and the fabric sees read-modify-write
// traffic (PMC round 1: 187 MB read / 147 MB written per launch against 98 / 44 algorithmic).  The buffer collects 32 columns
// per row and the wave flushes whole 128-byte lines (8 lanes x 16 bytes per row).
constexpr int kGlStride = 36;    // floats per buffered row: 16-byte aligned, rows 36 banks apart

// (exact general kernel at 32 rows per wave, H <= 128: two waves per SIMD are part of the design -- the register budget is held
// to 256)
// KS > 1 (pair-symmetric exact kernel only; small M): KS waves of the workgroup share one block of rows and split the periods
// (values of h) between them; each writes the Gx columns of its own h range, the dX partial sums meet in the LDS scratch
// (see cin_tail_dz_kernel).
// pair-symmetric exact kernel at 32 rows per wave: the slot counts whose instantiation fits 256 registers without spilling
constexpr bool cin_dzs_two_waves(int JT) { return JT <= 12 || JT == 16; }
template <int MB, int JT, int NHMAX, bool SYM = false, int KS = 1>
__global__ __launch_bounds__(256, (MB == 1 && NHMAX == 64 && KS == 1 && (!SYM || cin_dzs_two_waves(JT))) ? 2 : 1) void cin_dz3_kernel(const float* __restrict__ gT, int HS, const float* __restrict__ Wz,
                                                         const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                         const float* __restrict__ dPprev, int ldp, int K, float* __restrict__ GprevT,
                                                         int HSp, float* __restrict__ gx0T, float* __restrict__ dxT, int accumulate,
                                                         int M, int F, int Hp, int H, int periods, int FR) {
  // Per-lane scratch in LDS, laid out [mb][j][tid] (lane-minor -> conflict-free, immediate offsets): the lane's x
  // fragment and its dX accumulators.  Every address is touched by exactly one lane: no barriers, no atomics.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NCOL = 2 * NHMAX;
  constexpr int NQ = NHMAX / 4;
  constexpr int P = JT / gcd_c(16, JT);
  constexpr int HPP = 16 * P / JT;
  static_assert(KS == 1 || SYM, "the period split exists for the pair-symmetric kernel");
  const int tid = threadIdx.x, lane = tid & 63, wave = KS == 1 ? tid >> 6 : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int kpart = wave % KS;
  const int wrow0 = ((blockIdx.x * 4 + wave) / KS) * (32 * MB);
  if (KS == 1 && wrow0 >= M) return;
  const int pq_ = (periods + KS - 1) / KS;
  const int per_lo = KS == 1 ? 0 : min(periods, kpart * pq_);
  const int per_hi = KS == 1 ? periods : (wrow0 < M ? min(periods, per_lo + pq_) : per_lo);
  // SYM: x entry and dX accumulator of a field sit side by side, [mb][f][x | dX][kSymStride]: one run-time row address per slot, the
  // rest are compile-time offsets of the LDS instructions
  constexpr int kSymRow = 2 * kSymStride;
  float* xs = SYM ? smem + wave * 32 + r : smem + tid;                                    // xs[(mb*JT + j)*256]   | SYM: xs[(mb*FR + f)*kSymRow]
  float* dxs = SYM ? xs + kSymStride : smem + MB * JT * 256 + tid;                        // dxs[(mb*JT + j)*256]  | SYM: xs + kSymStride
  constexpr bool GLINE = !SYM;                                                             // (the first layer has no G^{l-1})
  float* gl = smem + 2 * MB * JT * 256 + wave * (MB * 32 * kGlStride);                     // GLINE: [mb][row 32][kGlStride]
  long mq[MB];
  bool vq[MB];
  float greg[MB][NHMAX], dpp[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wrow0 + mb * 32 + r;
    vq[mb] = m < M;
    mq[mb] = vq[mb] ? m : M - 1;
    if constexpr (SYM) {
      // eight loads per batch (unconditional, clamped, masked with an AND), then the LDS writes: the rolled form was one
      // load -> s_waitcnt vmcnt(0) -> ds_write per field, FR/2 exposed load latencies in a row per 32 rows
      for (int f0 = half; f0 < FR; f0 += 16) {
        float xt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xt[u] = xT[mq[mb] * F + min(f0 + 2 * u, F - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + 2 * u;
          if (f < FR) {
            const int keep = (vq[mb] && f < F) ? -1 : 0;
            xs[(mb * FR + f) * kSymRow] = __builtin_bit_cast(float, __builtin_bit_cast(int, xt[u]) & keep);
            dxs[(mb * FR + f) * kSymRow] = 0.f;
          }
        }
      }
    } else {
      // all JT loads first (unconditional, clamped, masked with an AND: see the per-period loads below), then the LDS writes:
      // a conditional load per element is JT exposed load latencies in a row before the wave's first MFMA
      float xt[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) xt[j] = xT[mq[mb] * F + min(2 * j + half, F - 1)];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int keep = (vq[mb] && 2 * j + half < F) ? -1 : 0;
        xs[(mb * JT + j) * 256] = __builtin_bit_cast(float, __builtin_bit_cast(int, xt[j]) & keep);
        dxs[(mb * JT + j) * 256] = 0.f;
      }
    }
    // the lane's half row of G: 16-byte loads (a dword per load made NHMAX requests per lane, and with every wave of the chip in
    // its prologue at once the L1 does not keep the lines between them: 4x the L2 requests of this form)
    const float4* grow4 = reinterpret_cast<const float4*>(gT + mq[mb] * HS + half * NHMAX);
#pragma unroll
    for (int s4 = 0; s4 < NHMAX / 4; ++s4) {
      const float4 g4 = grow4[s4];
      const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int keep = (vq[mb] && half * NHMAX + 4 * s4 + e < H) ? -1 : 0;
        greg[mb][4 * s4 + e] = __builtin_bit_cast(float, __builtin_bit_cast(int, gv[e]) & keep);
      }
    }
    dpp[mb] = 0.f;
    if (dPprev != nullptr && vq[mb]) {
      const long bb = mq[mb] / K;
      dpp[mb] = dPprev[bb * ldp + (mq[mb] - bb * K)];
    }
  }
  if constexpr (SYM) __builtin_amdgcn_wave_barrier();  // the halves of a row read each other's x entries from here on
  const float4* wz = reinterpret_cast<const float4*>(Wz) + ((long)r * NCOL + half * NHMAX) / 4;
  constexpr long kTileStride = 32L * NCOL / 4;  // float4 per tile
  float4 q[NQ];
#pragma unroll
  for (int s4 = 0; s4 < NQ; ++s4) q[s4] = wz[(long)per_lo * P * kTileStride + s4];
  float gx[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) gx[mb] = 0.f;
  // destination of the completed G^{l-1}[m,h] values (selected once: keeps the tile loop free of invariant branches)
  float* gdst[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) gdst[mb] = gx0T != nullptr ? gx0T + mq[mb] * F : GprevT + mq[mb] * HSp;

  // The 16 slots of tile t are contracted while the MFMA chain of tile t+1 runs (one slot per 4-step group), so the
  // VALU/LDS epilogue hides behind the matrix pipe: dprev/xprev/hprev describe the tile being contracted.
  f32x16 dprev[MB];
  float xprev[MB][HPP], xcur[MB][HPP];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dprev[mb][i] = 0.f;
#pragma unroll
    for (int hl = 0; hl < HPP; ++hl) xprev[mb][hl] = xcur[mb][hl] = 0.f;
  }
  int hprev = -HPP;  // h base of the period the previous tile belongs to (the fake tile before the first one stores nothing)
  float gout[MB][HPP];
  const bool gvec = gx0T == nullptr;  // G^{l-1} rows are 512-byte aligned; the layer-1 side buffer [M][F] is not

  // A slot is contracted in two halves one step group apart, so the LDS reads are back before they are needed (a
  // read-then-wait inside a group parks the wave, and with one wave per SIMD that starves the MFMA pipe):
  // slot_fetch reads the slot's x entry and dX accumulator, slot_apply (next group) does the two FMAs and the write.
  // Within a group the apply (write) precedes the next fetch (read), so slots that alias one word stay ordered.
  float lx[MB], ldx[MB];
  int lfo = 0;
  // SYM: row index (in floats) of field hb + half, per period: the slot's field is that row + (hl + 2j) rows, minus F rows past the
  // wrap -- a compare against a scalar, a select between the two precomputed bases, and a compile-time LDS offset per slot (the
  // form with the field computed per slot cost ~9 vector instructions and a 32-bit multiply each).  The fake tile before the first
  // one (hb < 0, dZ = 0) is clamped to h = 0.
  int symb = 0, symw = 0, symh = 0;
  auto sym_period = [&](int hb) {
    symh = max(hb, 0) + half;
    symb = symh * kSymRow;
    symw = symb - F * kSymRow;
  };
  auto slot_fetch = [&](int hb, int tp, int rr) {
    const int sp = 16 * tp + rr;
    const int hl = sp / JT, j = sp % JT;  // compile-time after unrolling
    if constexpr (SYM) {
      // f = (h + 2j + half) mod F; h + 2j + half < F + FR by the choice of FR
      (void)hb;
      lfo = (symh >= F - (hl + 2 * j) ? symw : symb) + (hl + 2 * j) * kSymRow;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int xi = SYM ? mb * FR * kSymRow + lfo : (mb * JT + j) * 256;
      lx[mb] = xs[xi];
      ldx[mb] = dxs[xi];
    }
  };
  auto slot_apply = [&](const f32x16 (&d)[MB], const float (&xpv)[MB][HPP], int hb, int tp, int rr) {
    const int sp = 16 * tp + rr;
    const int hl = sp / JT, j = sp % JT;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float dz = d[mb][rr];
      const int xi = SYM ? mb * FR * kSymRow + lfo : (mb * JT + j) * 256;
      gx[mb] = fmaf(dz, lx[mb], gx[mb]);
      dxs[xi] = fmaf(dz, xpv[mb][hl], ldx[mb]);
      if (j == JT - 1) {
        // h = hb + hl is complete: collect the period's HPP values and store them with one 16/8/4-byte access
        // (single-dword stores at a row stride turn into one partial-line write each: 7x write amplification)
        gout[mb][hl] = lane_halves_sum(gx[mb]) + dpp[mb];
        gx[mb] = 0.f;
        if (GLINE && gvec) {
          if (hl == HPP - 1 && half == 0 && hb >= 0) {
            float* bl = gl + (mb * 32 + r) * kGlStride + (hb & 31);
            if constexpr (HPP == 4) *reinterpret_cast<float4*>(bl) = make_float4(gout[mb][0], gout[mb][1], gout[mb][2], gout[mb][3]);
            else if constexpr (HPP == 2) *reinterpret_cast<float2*>(bl) = make_float2(gout[mb][0], gout[mb][1]);
            else bl[0] = gout[mb][0];
          }
        } else if (hl == HPP - 1 && half == 0 && vq[mb] && hb >= 0) {
          float* dst = gdst[mb] + hb;
          if (gvec && hb + HPP <= Hp) {
            if constexpr (HPP == 4) *reinterpret_cast<float4*>(dst) = make_float4(gout[mb][0], gout[mb][1], gout[mb][2], gout[mb][3]);
            else if constexpr (HPP == 2) *reinterpret_cast<float2*>(dst) = make_float2(gout[mb][0], gout[mb][1]);
            else dst[0] = gout[mb][0];
          } else {
#pragma unroll
            for (int u = 0; u < HPP; ++u)
              if (hb + u < Hp) dst[u] = gout[mb][u];
          }
        }
      }
    }
  };
  // whole lines of G^{l-1}: columns [32 line, 32 line + 32) of the wave's rows, 8 lanes x 16 bytes per row
  auto flush_line = [&](int line) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 64 + lane, row = idx >> 3, c4 = idx & 7;
        const int m = wrow0 + mb * 32 + row, col = 32 * line + 4 * c4;
        const float4 v = *reinterpret_cast<const float4*>(gl + (mb * 32 + row) * kGlStride + 4 * c4);
        if (m < M && col < Hp) {
          float* dst = GprevT + (long)m * HSp + col;
          if (col + 3 < Hp) *reinterpret_cast<float4*>(dst) = v;
          else {
            dst[0] = v.x;
            if (col + 1 < Hp) dst[1] = v.y;
            if (col + 2 < Hp) dst[2] = v.z;
          }
        }
      }
  };
  sym_period(hprev);
  slot_fetch(hprev, P - 1, 0);

#pragma unroll 1
  for (int per = per_lo; per < per_hi; ++per) {
    const int hbase = per * HPP;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int hl = 0; hl < HPP; ++hl) {
        // Unconditional load, masked with an AND.  A conditional load (and equally a select on a load, which the compiler
        // turns back into a branch around the load) costs a branch + s_waitcnt vmcnt(0) per value at the top of every period:
        // HPP exposed load latencies in a row, and the A-operand queue drained each time.
        const float xv = xpT[mq[mb] * xps + min(hbase + hl, Hp - 1)];
        const int keep = (vq[mb] && hbase + hl < Hp) ? -1 : 0;
        xcur[mb][hl] = __builtin_bit_cast(float, __builtin_bit_cast(int, xv) & keep);
      }
#pragma unroll
    for (int tp = 0; tp < P; ++tp) {
      // the stream is allocated one tile past the last period, so the refill never leaves the buffer
      const float4* wnext = wz + ((long)per * P + tp + 1) * kTileStride;
      f32x16 d[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[mb][i] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < NQ; ++s4) {
        const float4 w = q[s4];
        q[s4] = wnext[s4];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          d[mb] = mfma32(w.x, greg[mb][4 * s4 + 0], d[mb]);
          d[mb] = mfma32(w.y, greg[mb][4 * s4 + 1], d[mb]);
          d[mb] = mfma32(w.z, greg[mb][4 * s4 + 2], d[mb]);
          d[mb] = mfma32(w.w, greg[mb][4 * s4 + 3], d[mb]);
        }
        // previous tile's slots, spread over this tile's step groups (NQ is 16 or 32; 16 slots per tile)
        if (s4 < 16) {
          if (tp == 0) slot_apply(dprev, xprev, hprev, P - 1, s4);
          else slot_apply(dprev, xcur, hbase, tp - 1, s4);
          if (s4 < 15) {
            if (tp == 0) slot_fetch(hprev, P - 1, s4 + 1);
            else slot_fetch(hbase, tp - 1, s4 + 1);
          } else {
            if (tp == 0) sym_period(hbase);   // (from here on the slots belong to this period)
            slot_fetch(hbase, tp, 0);  // first slot of this tile, applied in the first group of the next one
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) dprev[mb] = d[mb];
      // the previous period's values became complete behind this period's first tile: flush when they close a line
      if constexpr (GLINE) {
        if (tp == 0 && gvec && hprev >= 0 && ((hprev + HPP) & 31) == 0) flush_line(hprev >> 5);
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int hl = 0; hl < HPP; ++hl) xprev[mb][hl] = xcur[mb][hl];
    hprev = hbase;
  }
  // the last tile's slots
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    slot_apply(dprev, xprev, hprev, P - 1, rr);
    if (rr < 15) slot_fetch(hprev, P - 1, rr + 1);
  }
  if constexpr (GLINE) {
    if (gvec && hprev >= 0) flush_line(hprev >> 5);   // the last (possibly partial) line
  }
  // dX rows of the wave are contiguous in dxT ([32*MB rows][F]): written cooperatively from the LDS scratch so that
  // every store instruction covers whole lines (one lane per row element = a 156-byte stride = 8-16x write
  // amplification in WRITE_SIZE)
  if constexpr (KS > 1) {
    __syncthreads();
    if (kpart > 0 || wrow0 >= M) return;
  } else {
    __builtin_amdgcn_wave_barrier();
  }
  const float* dsc = smem + (SYM ? kSymStride + wave * 32 : MB * JT * 256 + wave * 64);
#pragma unroll 1
  for (int mb = 0; mb < MB; ++mb) {
    const int row0 = wrow0 + mb * 32;
    const int nrow = min(32, M - row0);
    float* dst = dxT + (long)row0 * F;
    for (int idx = lane; idx < nrow * F; idx += 64) {
      const int rr = idx / F, f = idx - rr * F;
      float v = SYM ? dsc[(mb * FR + f) * kSymRow + rr] : dsc[(mb * JT + (f >> 1)) * 256 + (f & 1) * 32 + rr];
      if constexpr (KS > 1) {   // (SYM only) the group's partial sums, in wave order
#pragma unroll
        for (int k = 1; k < KS; ++k) v += dsc[(mb * FR + f) * kSymRow + rr + 32 * k];
      }
      dst[idx] = accumulate ? dst[idx] + v : v;
    }
  }
}

// Backward weight path, streaming form: dW[c,n] = sum_m Z[m,c] G[m,n].  Wave = 32*MB channel rows (on the lanes as
// A-operand rows) x one 128-column chunk; the reduction runs over m (two rows per step, one per wave half) with all
// three operands streamed from L2 through a DEPTH-step register queue:
//   A[c][m] = x^{l-1}[m,h_c] * x[m,f_c]   (two dword gathers: 32 consecutive f -> one 128-byte segment; <= 3 distinct h)
//   B[m][n] = G[m, 4r..4r+3]              (one aligned 16-byte load)
// xT == nullptr stands for a single all-ones field (used for the last layer's rank-one weight gradient).
constexpr int kDwDepth = 8;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The three operand streams are read with raw buffer loads: the per-lane part of every address is a constant 32-bit
// voffset, the per-step part a scalar soffset, and rows past the end of a tensor read as zero through the
// descriptor's range check -- no per-load 64-bit address arithmetic, no clamping (plain global loads with per-step
// 64-bit address math measured 105 TFLOP/s on this kernel, this form 124).

template <int MB, bool XONES, int DEPTH = kDwDepth>
__global__ __launch_bounds__(256, 1) void cin_dw3_kernel(const float* __restrict__ gT, int HS, const float* __restrict__ xT,
                                                          const float* __restrict__ xpT, int xps, float* __restrict__ part, int M, int F,
                                                          int Hp, int H, int rows_per_split, int blocks_x, int chunks, int items, int symD,
                                                          int xtra = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  // symD > 0 (first layer, x^{l-1} = x): channels are the unordered pairs c = h*symD + d <-> (h, f = (h+d) mod F)
  // xtra > 0 (quadratic tail): `xtra` more channels behind the pairs, c = F*symD + f <-> x^{l-1} column F (an extra column of the
  // xpT array: a per-row scale) times x[m,f] -- a second, rank-F product riding in the same pass over G
  const int C = symD > 0 ? F * symD + xtra : Hp * F;
  // XCD-aware work mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup i
  // of XCD i%8 takes item (i%8)*(grid/8) + i/8 of the split-major item list: an XCD then streams only its own
  // row splits of G / x / x^{l-1} through its L2 instead of all of them (grid is a multiple of 8).
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (item >= items) return;
  const int bx = item % blocks_x;
  const int chunk = (item / blocks_x) % chunks;
  const int split = item / (blocks_x * chunks);
  const int c0 = (bx * 4 + wave) * (32 * MB);
  if (c0 >= C) return;
  // descriptors start at the split's first row (offsets stay far below the 2 GiB descriptor range for any M) and end
  // with the tensor, so the prefetch past the last row reads zeros
  const int m_lo = split * rows_per_split;
  const int m_hi = min(M, m_lo + rows_per_split);
  // the descriptors cover exactly the split's rows [m_lo, m_hi): the last step's spare row and the prefetch past the end read
  // zeros, so the loop needs no row masks (0 * 0 products); channel rows past C compute finite values that are never stored
  const long mrem = (long)m_hi - m_lo;
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(gT + (long)m_lo * HS, mrem * HS * 4);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(XONES ? gT : xT + (long)m_lo * F, mrem * F * 4);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(xpT + (long)m_lo * xps, mrem * xps * 4);
  int fo[MB], ho[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int c = c0 + mb * 32 + r;
    const int cc = c < C ? c : C - 1;
    int hh = symD > 0 ? cc / symD : cc / F;
    int ff = symD > 0 ? (hh + (cc - hh * symD)) % F : cc - hh * F;
    if (symD > 0 && cc >= F * symD) {
      hh = F;
      ff = cc - F * symD;
    }
    ho[mb] = (half * xps + hh) * 4;            // byte offsets of the lane's column inside row (m_lo + half)
    fo[mb] = (half * F + ff) * 4;
  }
  const int go = (half * HS + chunk * 128 + 4 * r) * 4;
  const int steps = (m_hi - m_lo + 1) >> 1;
  const int groups = (steps + DEPTH - 1) / DEPTH;

  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  f32x4v qg[DEPTH];
  float qx[DEPTH][MB], qp[DEPTH][MB];
  // step s reads rows m_lo + 2s (+half): scalar byte offsets advance by two rows per step
  auto fetch = [&](int s, f32x4v& g4, float (&xv)[MB], float (&pv)[MB]) {
    const int row = 2 * s;  // uniform, relative to the split's first row
    g4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rg, go, row * HS * 4, 0));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if constexpr (XONES) xv[mb] = 1.f;
      else xv[mb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, fo[mb], row * F * 4, 0));
      pv[mb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, ho[mb], row * xps * 4, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, qg[d], qx[d], qp[d]);
  // the generated operand of step s+1 is computed in front of step s's MFMAs and consumed a step later (a v_mul feeding the very
  // next MFMA costs its VALU -> MFMA operand wait states on the matrix pipe, tools/probe_mfma16.hip)
  float ac[MB], an[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) ac[mb] = qx[0][mb] * qp[0][mb];
#pragma unroll 1
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int s = g * DEPTH + d;
      const f32x4v g4 = qg[d];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) an[mb] = qx[(d + 1) % DEPTH][mb] * qp[(d + 1) % DEPTH][mb];   // (slot 0: refilled with the next group's first step)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][0] = mfma32(ac[mb], g4[0], acc[mb][0]);
        acc[mb][1] = mfma32(ac[mb], g4[1], acc[mb][1]);
        acc[mb][2] = mfma32(ac[mb], g4[2], acc[mb][2]);
        acc[mb][3] = mfma32(ac[mb], g4[3], acc[mb][3]);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) ac[mb] = an[mb];
      __builtin_amdgcn_sched_barrier(0);
      // the refill comes AFTER the step's MFMAs so that it may land in the registers it replaces: loaded into fresh registers,
      // the eight slots have to be copied back at the end of every group, and those copies wait for every load of the group
      // (c4 l2, A/B on one box: 0.661 -> 0.617 ms).  Unlike the forward kernel, waiting for the prologue's loads before the
      // loop -- which lets the full 8-deep queue stay in flight at the top of every group -- made this kernel SLOWER (0.708 ms
      // with three resident waves per SIMD): the loop-top wait the compiler derives here (8 loads in flight) stays.
      fetch(s + DEPTH, qg[d], qx[d], qp[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* pout = part + (long)split * C * H;
  const bool vec = (H & 3) == 0;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int c = c0 + mb * 32 + mfma32_row(reg, half);
      if (c < C) {
        const int n = chunk * 128 + 4 * r;
        float* dst = pout + (long)c * H + n;
        if (vec && n + 3 < H) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[mb][0][reg], acc[mb][1][reg], acc[mb][2][reg], acc[mb][3][reg]);
        } else {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            if (n + nb < H) dst[nb] = acc[mb][nb][reg];
        }
      }
    }
  }
}

// part[blk][n] = sum over a chunk of rows of gT[m][n]: 16-byte loads, HS/4 threads per row and 256/(HS/4) row groups
// per workgroup (two rows in flight per thread), the groups folded through LDS in a fixed order.
static __global__ __launch_bounds__(256) void cin_colsum3_kernel(const float* __restrict__ gT, int HS, float* __restrict__ part, int M, int H,
                                                          int rows_per_block) {
  __shared__ float4 red[256];
  const int q = HS >> 2;  // float4 per row: 32 or 64
  const int c4 = threadIdx.x % q, g = threadIdx.x / q, ng = 256 / q;
  const long m_lo = (long)blockIdx.x * rows_per_block, m_hi = min((long)M, m_lo + rows_per_block);
  const float4* src = reinterpret_cast<const float4*>(gT) + c4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  long m = m_lo + g;
  for (; m + ng < m_hi; m += 2 * ng) {
    const float4 u = src[m * q], v = src[(m + ng) * q];
    a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
    a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
  }
  if (m < m_hi) {
    const float4 u = src[m * q];
    a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
  }
  red[threadIdx.x] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (g == 0) {
    float4 t = red[c4];
    for (int k = 1; k < ng; ++k) {
      const float4 u = red[k * q + c4];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float* dst = part + (long)blockIdx.x * H + 4 * c4;
    if (4 * c4 < H) dst[0] = t.x;
    if (4 * c4 + 1 < H) dst[1] = t.y;
    if (4 * c4 + 2 < H) dst[2] = t.z;
    if (4 * c4 + 3 < H) dst[3] = t.w;
  }
}

// gT[m][n] = dP[b*ldp + k] for n < H (general path for the top layer: the pooled gradient broadcast over feature maps)
static __global__ __launch_bounds__(256) void cin_bcast3_kernel(const float* __restrict__ dP, int ldp, int K, float* __restrict__ gT, int HS,
                                                         int M, int H) {
  const long total = (long)M * HS;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / HS;
    const int n = (int)(i - m * HS);
    const long b = m / K;
    gT[i] = n < H ? dP[b * ldp + (m - b * K)] : 0.f;
  }
}

// yT[m][0..YS) = xT[m][f] * dP[m] for f < F, zero beyond (right-hand side of the last layer's rank-one dW); YS = F
// rounded up to 4 keeps the rows 16-byte aligned for the dW kernel's B-operand loads (its lanes past YS/4 read into the
// following rows: finite values feeding columns that are never stored)
static __global__ __launch_bounds__(256) void cin_scale_rows3_kernel(const float* __restrict__ xT, const float* __restrict__ dP, int ldp, int K,
                                                              float* __restrict__ yT, int M, int F, int YS) {
  const long total = (long)M * YS;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / YS;
    const int f = (int)(i - m * YS);
    const long b = m / K;
    yT[i] = f < F ? xT[m * F + f] * dP[b * ldp + (m - b * K)] : 0.f;
  }
}

// out[i] = sum_{p < parts} part[p*n + i]   (fixed order).  One workgroup per 64 outputs; the 4 waves take every 4th
// partial (coalesced over i), then the 4 wave sums are added in wave order -> many loads in flight, fixed order.
// out2 != nullptr: outputs i >= n1 go to out2[i - n1] (the dense head: ddense_w | ddense_b from one partial buffer, no copies).
// pstride: distance between consecutive partials (0 = n: densely packed)
__device__ __forceinline__ void cin_reduce_body(const float* __restrict__ part, float* __restrict__ out, long n, int parts,
                                                float* __restrict__ out2, long n1, int bid, long pstride = 0) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i = (long)bid * 64 + lane;
  const long ps = pstride > 0 ? pstride : n;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < n) {
    int p = wave;
    for (; p + 60 < parts; p += 64) {   // (sixteen loads in flight per lane, summed as the loop below does: with hundreds of partials
      float u[16];                      // four at a time was the longest dependent chain of the launches this rides in)
#pragma unroll
      for (int e = 0; e < 16; ++e) u[e] = part[(long)(p + 4 * e) * ps + i];
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        t0 += u[e];
        t1 += u[e + 1];
        t2 += u[e + 2];
        t3 += u[e + 3];
      }
    }
    for (; p + 12 < parts; p += 16) {
      t0 += part[(long)p * ps + i];
      t1 += part[(long)(p + 4) * ps + i];
      t2 += part[(long)(p + 8) * ps + i];
      t3 += part[(long)(p + 12) * ps + i];
    }
    for (; p < parts; p += 4) t0 += part[(long)p * ps + i];
  }
  red[wave][lane] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (wave == 0 && i < n) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (out2 != nullptr && i >= n1) out2[i - n1] = v;
    else out[i] = v;
  }
}
static __global__ __launch_bounds__(256) void cin_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int parts, float* __restrict__ out2 = nullptr, long n1 = 0, long pstride = 0) {
  cin_reduce_body(part, out, n, parts, out2, n1, blockIdx.x, pstride);
}

// Pair-indexed first-layer weight gradient: fixed-order sum of the row-split partials (as cin_reduce_kernel: 64 outputs per
// workgroup, the 4 waves take every 4th partial) written straight to BOTH dW rows of the pair (cin_expand_sym_kernel's map):
// slot (h, d) <-> f = (h + d) mod F gives dW[(h,f)] and, unless d == 0 or 2d == F (that pair's other end has its own slot),
// dW[(f,h)].
// vout != nullptr: the workgroups past the pair rows reduce nv more values that sit behind them in every partial (the extra
// channel rows of cin_dw3_kernel's xtra) into vout, unexpanded
static __global__ __launch_bounds__(256) void cin_reduce_expand_sym_kernel(const float* __restrict__ part, float* __restrict__ dW, int F, int D,
                                                                    int H, int parts, long pstride = 0, float* __restrict__ vout = nullptr,
                                                                    long nv = 0) {
  if (vout != nullptr) {
    const long nbw = ((long)F * D * H + 63) / 64;
    if ((long)blockIdx.x >= nbw) {
      cin_reduce_body(part + (long)F * D * H, vout, nv, parts, nullptr, 0, (int)(blockIdx.x - nbw), pstride);
      return;
    }
  }
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n = (long)F * D * H;
  const long ps = pstride > 0 ? pstride : n;   // (partials that carry extra rows behind the pair rows: cin_dw3_kernel's xtra)
  const long i = (long)blockIdx.x * 64 + lane;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < n) {
    int p = wave;
    for (; p + 60 < parts; p += 64) {   // (sixteen loads in flight per lane, summed as the loop below does: with hundreds of partials
      float u[16];                      // four at a time was the longest dependent chain of the launches this rides in)
#pragma unroll
      for (int e = 0; e < 16; ++e) u[e] = part[(long)(p + 4 * e) * ps + i];
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        t0 += u[e];
        t1 += u[e + 1];
        t2 += u[e + 2];
        t3 += u[e + 3];
      }
    }
    for (; p + 12 < parts; p += 16) {
      t0 += part[(long)p * ps + i];
      t1 += part[(long)(p + 4) * ps + i];
      t2 += part[(long)(p + 8) * ps + i];
      t3 += part[(long)(p + 12) * ps + i];
    }
    for (; p < parts; p += 4) t0 += part[(long)p * ps + i];
  }
  red[wave][lane] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (wave == 0 && i < n) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int col = (int)(i % H);
    const int pair = (int)(i / H);
    const int h = pair / D, d = pair - h * D;
    const int f = (h + d) % F;
    dW[((long)h * F + f) * H + col] = v;
    if (d != 0 && 2 * d != F) dW[((long)f * F + h) * H + col] = v;
  }
}

// part[blk] = sum over a chunk of samples of dP[b*ldp + k], k < K  (dbias of the last layer: same value for every n)
static __global__ __launch_bounds__(256) void cin_slice_sum_kernel(const float* __restrict__ dP, int ldp, float* __restrict__ part, int B,
                                                            int K, int bchunk) {
  __shared__ float red[256];
  const int b_lo = blockIdx.x * bchunk, b_hi = min(B, b_lo + bchunk);
  float t = 0.f;
  const int total = (b_hi - b_lo) * K;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int b = b_lo + i / K, k = i % K;
    t += dP[(long)b * ldp + k];
  }
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// =================================================================================================
// Last-layer shortcut.  The last feature map x^L is only ever sum-pooled over its feature-map axis n
// (reference :322), so with wsum[c] = sum_n W_L[c,n]:
//   p_L[m]            = sum_c Z[m,c] wsum[c] + sum_n bias[n]
//   G^L[m,n]          = dP_L[m] for every n  =>  dZ_L[m,c] = dP_L[m] wsum[c],  dW_L[c,n] = sum_m Z[m,c] dP_L[m] (all n),
//                       dbias_L[n] = sum_m dP_L[m]
// i.e. two [M x F] x [F x Hp] / [M x Hp] x [Hp x F] products instead of [M x Hp*F] x [Hp*F x H] GEMMs: 1/H of the
// flops, run by small VALU kernels (one thread per row m, wsum broadcast from LDS).  Results are those of the
// general kernels up to fp32 rounding; fil_cin_* mode 1 forces the general path for validation.
constexpr int kLastFMax = 64;  // F <= 64

static __global__ __launch_bounds__(256) void cin_wsum_kernel(const float* __restrict__ W, float* __restrict__ wsum, int C, int H) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  float t = 0.f;
  if (row < C)
    for (int n = l; n < H; n += 32) t += W[(long)row * H + n];
  t = half_wave_sum(t);
  if (row < C && l == 0) wsum[row] = t;
}

// wsum and, when wsn != nullptr, its copy in the forward kernel's operand layout, wsn[chunk][fp < JT2][128] = wsum[(chunk*128 + col)*F + fp], from ONE launch: the
// workgroup that sums row (n, fp) of W also stores it at wsn[chunk][fp][col]; the padding of wsn (fp >= F, n >= Hp) is zeroed
// by a grid-stride pass over its index space.  (A kernel this small costs its ~4.5 us of dispatch, not its work.)
static __global__ __launch_bounds__(256) void cin_wsum_wsn_kernel(const float* __restrict__ W, float* __restrict__ wsum, int C, int H,
                                                           float* __restrict__ wsn, int Hp, int F, int JT2, int chunks) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  float t = 0.f;
  if (row < C)
    for (int n = l; n < H; n += 32) t += W[(long)row * H + n];
  t = half_wave_sum(t);
  if (row < C && l == 0) {
    wsum[row] = t;
    if (wsn != nullptr) {
      const int n = row / F, fp = row - n * F;          // row = n*F + fp (n < Hp, fp < F)
      wsn[(((n >> 7) * JT2) + fp) * 128 + (n & 127)] = t;
    }
  }
  if (wsn != nullptr) {
    const int total = chunks * JT2 * 128;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const int col = i & 127, fp = (i >> 7) % JT2, chunk = (i >> 7) / JT2;
      if (fp >= F || chunk * 128 + col >= Hp) wsn[i] = 0.f;
    }
  }
}

// stages wsum as [Hp][FP4] (FP4 = F rounded up to 4, zero padded) so rows can be read 16 bytes at a time
__device__ __forceinline__ void stage_wsum(const float* __restrict__ wsum, float* ws, int Hp, int F, int FP4) {
  for (int idx = threadIdx.x; idx < Hp * FP4; idx += blockDim.x) {
    const int h = idx / FP4, f = idx - h * FP4;
    ws[idx] = f < F ? wsum[h * F + f] : 0.f;
  }
}

// One row m is shared by 4 lanes (hq = lane>>4 takes h = hq, hq+4, ...): a wave covers 16 consecutive rows (k
// contiguous -> 64-byte segments of x^{L-1}[b,h,:]), a workgroup 64 rows; partial sums are folded with two shuffles.
constexpr int kLastRows = 64;

static __global__ __launch_bounds__(256) void cin_last_fwd_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                           const float* __restrict__ wsum, const float* __restrict__ bias,
                                                           float* __restrict__ pool, int M, int F, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int FP4 = (F + 3) & ~3;
  stage_wsum(wsum, smem, Hp, F, FP4);
  float bsum = 0.f;
  for (int n = 0; n < H; ++n) bsum += bias[n];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * kLastRows + wave * 16 + (lane & 15), hq = lane >> 4;
  const bool valid = m < M;
  const long mm = valid ? m : 0;
  float xr[kLastFMax];
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) xr[f] = (valid && f < F) ? xT[mm * F + f] : 0.f;
  const float* xprow = xpT + mm * xps;
  float p = 0.f;
  for (int h = hq; h < Hp; h += 4) {
    const float4* wrow = reinterpret_cast<const float4*>(smem + h * FP4);
    float t = 0.f;
#pragma unroll
    for (int f4 = 0; f4 < kLastFMax / 4; ++f4) {
      if (4 * f4 < F) {
        const float4 w = wrow[f4];
        t = fmaf(xr[4 * f4], w.x, t);
        t = fmaf(xr[4 * f4 + 1], w.y, t);
        t = fmaf(xr[4 * f4 + 2], w.z, t);
        t = fmaf(xr[4 * f4 + 3], w.w, t);
      }
    }
    p = fmaf(valid ? xprow[h] : 0.f, t, p);
  }
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  if (valid && hq == 0) pool[m] = p + bsum;
}

// Gprev[m,h] = dP[m] * sum_f x[m,f] wsum[h,f] (+ dPprev[m]);  dX[m,f] = dP[m] * sum_h x^{L-1}[m,h] wsum[h,f]
// layer1 (L == 1, x^{L-1} == x): both terms go to dX (the first one through a small LDS tile).
static __global__ __launch_bounds__(256) void cin_last_bwd_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                           const float* __restrict__ wsum, const float* __restrict__ dP, int ldp,
                                                           const float* __restrict__ dPprev, float* __restrict__ GprevT, int HSp,
                                                           float* __restrict__ dxT, int layer1, int M, int F, int K, int Hp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int FP4 = (F + 3) & ~3;
  float* ts = smem + Hp * FP4;  // [kLastRows][kLastFMax + 1], layer1 only
  stage_wsum(wsum, smem, Hp, F, FP4);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = wave * 16 + (lane & 15), hq = lane >> 4;
  const int m = blockIdx.x * kLastRows + rowl;
  const bool valid = m < M;
  const long mm = valid ? m : 0;
  const int b = (int)(mm / K), k = (int)(mm - (long)b * K);
  float xr[kLastFMax], u[kLastFMax];
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) {
    xr[f] = (valid && f < F) ? xT[mm * F + f] : 0.f;
    u[f] = 0.f;
  }
  const float dp = valid ? dP[(long)b * ldp + k] : 0.f;
  const float dpp = (valid && dPprev != nullptr) ? dPprev[(long)b * ldp + k] : 0.f;
  const float* xprow = xpT + mm * xps;
  for (int h = hq; h < Hp; h += 4) {
    const float4* wrow = reinterpret_cast<const float4*>(smem + h * FP4);
    const float xph = valid ? xprow[h] : 0.f;
    float t = 0.f;
#pragma unroll
    for (int f4 = 0; f4 < kLastFMax / 4; ++f4) {
      if (4 * f4 < F) {
        const float4 w = wrow[f4];
        t = fmaf(xr[4 * f4], w.x, t);
        t = fmaf(xr[4 * f4 + 1], w.y, t);
        t = fmaf(xr[4 * f4 + 2], w.z, t);
        t = fmaf(xr[4 * f4 + 3], w.w, t);
        u[4 * f4] = fmaf(xph, w.x, u[4 * f4]);
        u[4 * f4 + 1] = fmaf(xph, w.y, u[4 * f4 + 1]);
        u[4 * f4 + 2] = fmaf(xph, w.z, u[4 * f4 + 2]);
        u[4 * f4 + 3] = fmaf(xph, w.w, u[4 * f4 + 3]);
      }
    }
    if (layer1) ts[rowl * (kLastFMax + 1) + h] = t;
    else if (valid) GprevT[mm * HSp + h] = fmaf(dp, t, dpp);
  }
  if (layer1) __syncthreads();
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) {
    if (f < F) {
      float v = u[f];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (layer1) v += ts[rowl * (kLastFMax + 1) + f];  // Hp == F: the x^{0} role of x
      if (valid && hq == 0) dxT[mm * F + f] = dp * v;
    }
  }
}

// dW[c,n] = v[c] for every n;  Ft > 0: v arrives transposed, v^T [Ft][Hp], c = h*Ft + f
// bpart != nullptr: workgroup 0 also finishes the layer's dbias (the same value for every n: the sum of the `bparts` slice
// partials, folded through LDS in a fixed order) -- one launch less
static __global__ __launch_bounds__(256) void cin_fill_rows_kernel(const float* __restrict__ v, float* __restrict__ dW, long C, int H, int Ft,
                                                            int Hp, const float* __restrict__ bpart = nullptr, int bparts = 0,
                                                            float* __restrict__ dbias = nullptr) {
  if (bpart != nullptr && blockIdx.x == 0) {
    __shared__ float red[256];
    float t = 0.f;
    for (int p = threadIdx.x; p < bparts; p += 256) t += bpart[p];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    const float tot = red[0];
    for (int n = threadIdx.x; n < H; n += 256) dbias[n] = tot;
  }
  const long total = C * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long c = i / H;
    dW[i] = Ft > 0 ? v[(c % Ft) * Hp + c / Ft] : v[c];
  }
}

// MFMA form of the last layer's data gradients (mode 0, L >= 2; the VALU kernel above stays for L == 1):
//   t[m,n] = sum_f x[m,f] wsum[n,f]         -> G^{L-1}[m,n] = dP[m] t[m,n] + dPprev[m]     (A = x fragment, B = wsn)
//   u[m,f] = sum_h x^{L-1}[m,h] wsum[h,f]   -> dX[m,f]      = dP[m] u[m,f]                 (A = the lane's x^{L-1} row)
// Wave = 32 rows.  wsn is wsum in the forward kernel's operand layout (cin_wsum_wsn_kernel) so lane r owns columns
// 4r..4r+3 of a chunk and G^{L-1} leaves as 16-byte stores; for u the reduction order is permuted (free in a GEMM)
// so that wave half `half` takes h = 4q + 2*half + {0,1}: one 8-byte load of its own row per two steps.
constexpr int kLast2Ld = 66;                  // floats between staged rows of x^{L-1} (64 columns + 2)
constexpr int kLast2Stage = 32 * kLast2Ld;    // per-wave staging floats: 32 rows x 64 columns of x^{L-1}, or the 32 rows of x (F <= 64)
template <int JT>
__global__ __launch_bounds__(256, 2) void cin_last_bwd2_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                            const float* __restrict__ wsum, const float* __restrict__ wsn,
                                                            const float* __restrict__ dP, int ldp, const float* __restrict__ dPprev,
                                                            float* __restrict__ GprevT, int HSp, float* __restrict__ dxT, int M, int F,
                                                            int K, int Hp, const float* __restrict__ Radd = nullptr, int HSr = 0,
                                                            const float* __restrict__ dPadd = nullptr, float* __restrict__ colpart = nullptr) {
  // Radd != nullptr (quadratic tail): G^{L-1}[m,n] += dPadd[m] * Radd[m,n] on the way out (one pass over G less)
  // colpart != nullptr: colpart[blockIdx][n < Hp] = sum over this workgroup's 128 rows of the G^{L-1} it writes (the partial
  // column sums of the next layer's dbias: cin_colsum3_kernel's output without its pass over G)
  __shared__ float colred[4][128];
  extern __shared__ __attribute__((aligned(16))) float smem[];  // wsum [Hp][F] | per wave: staging [kLast2Stage]
  for (int i = threadIdx.x; i < Hp * F; i += 256) smem[i] = wsum[i];
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M && colpart == nullptr) return;   // (with column sums every wave stays for the workgroup barriers; its rows are masked)
  const long mq = min(wrow0 + r, M - 1);
  // A lane needs ITS ROW of x (and, below, of x^{L-1}): read straight from memory that is one load instruction per element with 64
  // lanes in 64 different lines (20 + 32 such loads per wave kept the vector memory pipeline busier than the bytes they moved).
  // The wave's 32 rows are contiguous in memory: they come in through LDS by coalesced 16-byte loads, the lanes read their rows there.
  float* stg = smem + ((Hp * F + 3) & ~3) + wave * kLast2Stage;
  const int nrow = min(32, M - wrow0);   // (<= 0: a wave past the end that only stays for the workgroup barriers)
  if (nrow > 0) {
    const float4* src = reinterpret_cast<const float4*>(xT + (long)wrow0 * F);   // 32 rows x F floats = a multiple of 16 bytes from an aligned start
    const int n4 = nrow * F / 4, tail = nrow * F - 4 * n4;
    for (int i = lane; i < n4; i += 64) *reinterpret_cast<float4*>(stg + 4 * i) = src[i];
    if (lane < tail) stg[4 * n4 + lane] = xT[(long)wrow0 * F + 4 * n4 + lane];
  }
  __builtin_amdgcn_wave_barrier();
  float xr[JT];
  {
    const float* xrow = stg + min(r, max(nrow, 1) - 1) * F;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int f = 2 * j + half;
      xr[j] = (f < F && nrow > 0) ? xrow[min(f, F - 1)] : 0.f;
    }
  }
  // the three pooled gradients of the wave's rows: lane r fetches row r's once, the store loops read them from LDS by row (48
  // registers per lane in the accumulator layout: the kernel sat at the 256-register limit with spills)
  __shared__ float dps[4][3][32];
  if (half == 0) {
    const int b = (int)(mq / K), k = (int)(mq - (long)b * K);
    dps[wave][0][r] = dP[(long)b * ldp + k];
    dps[wave][1][r] = dPprev != nullptr ? dPprev[(long)b * ldp + k] : 0.f;
    dps[wave][2][r] = Radd != nullptr ? dPadd[(long)b * ldp + k] : 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  auto phase_t = [&]() {
  const int chunks = (Hp + 127) >> 7;
  for (int chunk = 0; chunk < chunks; ++chunk) {
    f32x16 t[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) t[nb][i] = 0.f;
    const float4* wsb = reinterpret_cast<const float4*>(wsn + (long)chunk * (2 * JT) * 128) + (half * 32 + r);
    // operand loads in batches of QB ahead of their MFMAs (load -> wait -> 4 MFMAs per step was JT exposed L2 latencies in a
    // row: the ISA had `s_waitcnt vmcnt(0)` in front of every step; the barrier keeps the scheduler from sinking them back)
    constexpr int QB = JT % 5 == 0 ? 5 : 4;
#pragma unroll
    for (int j0 = 0; j0 < JT; j0 += QB) {
      float4 wq[QB];
#pragma unroll
      for (int j = 0; j < QB; ++j) wq[j] = wsb[(long)(2 * (j0 + j)) * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const float4 w = wq[j];
        t[0] = mfma32(xr[j0 + j], w.x, t[0]);
        t[1] = mfma32(xr[j0 + j], w.y, t[1]);
        t[2] = mfma32(xr[j0 + j], w.z, t[2]);
        t[3] = mfma32(xr[j0 + j], w.w, t[3]);
      }
    }
    const int n0 = chunk * 128 + 4 * r;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    // the rows of R this lane adds on the way out, eight loads in flight at a time (one by one inside the loop they were 16 exposed
    // memory latencies in a row per wave)
    float4 rq[8];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      if (Radd != nullptr && (reg & 7) == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long mm = min(wrow0 + mfma32_row(reg + u, half), M - 1);
          rq[u] = *reinterpret_cast<const float4*>(Radd + mm * HSr + n0);   // (rows of a feature map are 512-byte aligned and padded to whole chunks)
        }
      }
      const int row = mfma32_row(reg, half);
      const int m = wrow0 + row;
      const float dpr = dps[wave][0][row], dppr = dps[wave][1][row], dpa = dps[wave][2][row];
      if (m < M) {
        float* dst = GprevT + (long)m * HSp + n0;
        float v0 = fmaf(dpr, t[0][reg], dppr), v1 = fmaf(dpr, t[1][reg], dppr);
        float v2 = fmaf(dpr, t[2][reg], dppr), v3 = fmaf(dpr, t[3][reg], dppr);
        if (Radd != nullptr) {
          const float4 ra = rq[reg & 7];
          v0 = fmaf(dpa, ra.x, v0);
          v1 = fmaf(dpa, ra.y, v1);
          v2 = fmaf(dpa, ra.z, v2);
          v3 = fmaf(dpa, ra.w, v3);
        }
        c0 += v0;
        c1 += v1;
        c2 += v2;
        c3 += v3;
        if (n0 + 3 < Hp) {
          *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
        } else {
          if (n0 < Hp) dst[0] = v0;
          if (n0 + 1 < Hp) dst[1] = v1;
          if (n0 + 2 < Hp) dst[2] = v2;
        }
      }
    }
    if (colpart != nullptr) {   // rows in register order, the two wave halves, then the four waves: a fixed order
      c0 = lane_halves_sum(c0);
      c1 = lane_halves_sum(c1);
      c2 = lane_halves_sum(c2);
      c3 = lane_halves_sum(c3);
      if (half == 0) *reinterpret_cast<float4*>(&colred[wave][4 * r]) = make_float4(c0, c1, c2, c3);
      __syncthreads();
      for (int n = tid; n < 128; n += 256)
        if (chunk * 128 + n < Hp) colpart[(long)blockIdx.x * Hp + chunk * 128 + n] = (colred[0][n] + colred[1][n]) + (colred[2][n] + colred[3][n]);
      __syncthreads();
    }
  }
  };
  auto phase_u = [&]() {
  if (wrow0 >= M) return;
  // u = x^{L-1} wsum: two column blocks of fields (f = r, f = 32 + r)
  f32x16 u0, u1;
#pragma unroll
  for (int i = 0; i < 16; ++i) u0[i] = u1[i] = 0.f;
  const bool two = F > 32;
  const int quads = (Hp + 3) >> 2;
#pragma unroll 1
  for (int q0 = 0; q0 < quads; q0 += 16) {
    // 64 columns of the wave's 32 rows of x^{L-1} through LDS: eight coalesced 16-byte loads per lane (a quarter row per 16 lanes),
    // rows kLast2Ld floats apart (8-byte aligned, conflict-free for the 8-byte row reads); columns >= Hp of a feature map are zero
    __builtin_amdgcn_wave_barrier();   // (the previous reads of the staging area are done)
    {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (lane >> 4) + 4 * i, c4 = lane & 15;
        const long mm = min(wrow0 + row, M - 1);
        v[i] = 4 * q0 + 4 * c4 < xps ? *reinterpret_cast<const float4*>(xpT + mm * xps + 4 * q0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (lane >> 4) + 4 * i, c4 = lane & 15;
        float* d = stg + row * kLast2Ld + 4 * c4;
        *reinterpret_cast<float2*>(d) = make_float2(v[i].x, v[i].y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(v[i].z, v[i].w);
      }
    }
    __builtin_amdgcn_wave_barrier();
    float2 av[16];
    const float* xprow = stg + r * kLast2Ld + 2 * half;
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = q0 + i < quads ? *reinterpret_cast<const float2*>(xprow + 4 * i) : make_float2(0.f, 0.f);
    // B operands (LDS) four steps at a time ahead of their MFMAs; out-of-range rows / fields read a clamped word and are
    // masked with an AND (read -> wait -> MFMA per step exposed the LDS latency at every step)
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {
      float b0[4][2], b1[4][2];
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int h = 4 * (q0 + i0 + ii) + 2 * half;
        const float* w0 = smem + min(h, Hp - 1) * F;
        const float* w1 = smem + min(h + 1, Hp - 1) * F;
        const int k0 = (h < Hp && r < F) ? -1 : 0, k1 = (h + 1 < Hp && r < F) ? -1 : 0;
        b0[ii][0] = __builtin_bit_cast(float, __builtin_bit_cast(int, w0[min(r, F - 1)]) & k0);
        b0[ii][1] = __builtin_bit_cast(float, __builtin_bit_cast(int, w1[min(r, F - 1)]) & k1);
        b1[ii][0] = b1[ii][1] = 0.f;
        if (two) {
          const int k2 = (h < Hp && r + 32 < F) ? -1 : 0, k3 = (h + 1 < Hp && r + 32 < F) ? -1 : 0;
          b1[ii][0] = __builtin_bit_cast(float, __builtin_bit_cast(int, w0[min(r + 32, F - 1)]) & k2);
          b1[ii][1] = __builtin_bit_cast(float, __builtin_bit_cast(int, w1[min(r + 32, F - 1)]) & k3);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        u0 = mfma32(av[i0 + ii].x, b0[ii][0], u0);
        u0 = mfma32(av[i0 + ii].y, b0[ii][1], u0);
        if (two) {
          u1 = mfma32(av[i0 + ii].x, b1[ii][0], u1);
          u1 = mfma32(av[i0 + ii].y, b1[ii][1], u1);
        }
      }
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int m = wrow0 + mfma32_row(reg, half);
    if (m < M) {
      const float dpr = dps[wave][0][mfma32_row(reg, half)];
      if (r < F) dxT[(long)m * F + r] = dpr * u0[reg];
      if (two && r + 32 < F) dxT[(long)m * F + 32 + r] = dpr * u1[reg];
    }
  }
  };
  // The two phases are independent (t: x, wsum, R -> G^{L-1}; u: x^{L-1}, wsum -> dX) and of different kinds -- t is mostly memory
  // traffic around 4 JT MFMAs, u mostly its 128 MFMAs.  Every other group of eight workgroups (neighbours on an XCD, likely the two
  // resident on one CU) runs them in the other order, so the waves sharing a SIMD are not both in the MFMA-bound phase at once:
  // 38.8 -> 35.4 us (c4, one box).  The results do not depend on the order.
  if ((blockIdx.x >> 3) & 1) {
    phase_u();
    phase_t();
  } else {
    phase_t();
    phase_u();
  }
}

struct PoolArgs {
  const float* part[kCinMaxL];  // [chunks][M] per layer
  int chunks[kCinMaxL];
};

// pooled[b, l*K+k] = sum_chunk part_l[chunk][b*K+k];  out[b] = pooled[b,:] . dense_w + dense_b
// One thread per pooled element, kHeadSamples samples per workgroup; the per-sample dot is summed in index order.
constexpr int kHeadSamples = 8;
static __global__ __launch_bounds__(256) void cin_head_fwd_kernel(PoolArgs pa, const float* __restrict__ dense_w,
                                                           const float* __restrict__ dense_b, float* __restrict__ pooled,
                                                           float* __restrict__ out, int B, int K, int L) {
  extern __shared__ float sh[];  // [kHeadSamples][L*K] products
  const int LK = L * K;
  const long M = (long)B * K;
  const int b0 = blockIdx.x * kHeadSamples;
  const int nb = min(kHeadSamples, B - b0);
  for (int e = threadIdx.x; e < nb * LK; e += 256) {
    const int bl = e / LK, j = e - bl * LK;
    const int l = j / K, k = j - l * K;
    const long b = b0 + bl;
    float v = 0.f;
    for (int ch = 0; ch < pa.chunks[l]; ++ch) v += pa.part[l][(long)ch * M + b * K + k];
    pooled[b * LK + j] = v;
    if (out != nullptr) sh[e] = v * dense_w[j];
  }
  if (out == nullptr) return;
  __syncthreads();
  if (threadIdx.x < nb) {
    float o = 0.f;
    for (int j = 0; j < LK; ++j) o += sh[threadIdx.x * LK + j];
    out[b0 + threadIdx.x] = o + dense_b[0];
  }
}

// output_dim == 1: dP[b,j] = g[b] * dense_w[j];  partial[blk][j] = sum_{b in blk} g[b]*pooled[b,j]  (j < LK),
// partial[blk][LK] = sum g[b].   Lane <-> j (LK + 1 <= 64 lanes... up to 256: see host check), the 256/64 waves of a
// workgroup split its chunk of samples and are folded in wave order.
static __global__ __launch_bounds__(256) void cin_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dense_w,
                                                           const float* __restrict__ pooled, float* __restrict__ dP,
                                                           float* __restrict__ part, int B, int LK, int bchunk) {
  __shared__ float red[4][256];
  const int b_lo = blockIdx.x * bchunk, b_hi = min(B, b_lo + bchunk);
  if (LK + 1 <= 64) {
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float t = 0.f;
    if (j <= LK) {
      const float wj = j < LK ? dense_w[j] : 0.f;
      for (int b = b_lo + w; b < b_hi; b += 4) {
        const float gb = g[b];
        if (j < LK) {
          dP[(long)b * LK + j] = gb * wj;
          t = fmaf(gb, pooled[(long)b * LK + j], t);
        } else {
          t += gb;
        }
      }
    }
    red[w][j] = t;
    __syncthreads();
    if (w == 0 && j <= LK) part[(long)blockIdx.x * (LK + 1) + j] = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
  } else {
    // wide heads (L*K + 1 > 64): a thread per column, looping when there are more columns than threads
    for (int j = threadIdx.x; j <= LK; j += 256) {
      const float wj = j < LK ? dense_w[j] : 0.f;
      float t = 0.f;
      for (int b = b_lo; b < b_hi; ++b) {
        const float gb = g[b];
        if (j < LK) {
          dP[(long)b * LK + j] = gb * wj;
          t = fmaf(gb, pooled[(long)b * LK + j], t);
        } else {
          t += gb;
        }
      }
      part[(long)blockIdx.x * (LK + 1) + j] = t;
    }
  }
}


}  // namespace fil
