// A3  xDeepFM CIN, "fused tail": the last TWO layers of the network contracted through the pooled weights of the last one.
//
// The reference (interactive_layer.py:316-323) only ever SUM-POOLS a feature map over its feature-map axis n, and feeds it to
// the next layer's 1x1 convolution.  For the last layer L that makes only wsum_L[(n,f)] = sum_n' W_L[(n,f),n'] observable
// (the round-1 "last-layer shortcut").  The same reasoning one level down: the map of the layer below, x^p (p = L-1, H_p
// columns), is used in exactly two ways --
//     pool_p[m] = sum_n x^p[m,n]                                       (its own sum-pool)
//     pool_L[m] = sum_f x[m,f] * (sum_n x^p[m,n] wsum_L[(n,f)])        (the last layer through wsum_L)
// i.e. only through the J = F+1 linear functionals  Bm[0][n] = 1,  Bm[1+f][n] = wsum_L[(n,f)]  of its columns.  With
//     Ueff[c][j] = sum_n W_p[c][n] Bm[j][n]      [C_p, F+1]      (a 51 MFLOP product at the north-star shape, per step)
// the two layers become ONE implicit GEMM with F+1 = 40 output columns instead of H_p = 128:
//     Y[m][j]  = sum_c Z_p[m,c] Ueff[c][j] + beff[j]                   pool_p = Y[:,0],  pool_L = sum_f x[:,f] Y[:,1+f] + sum bias_L
// and in the backward, with A[m][0] = dP_p[m], A[m][1+f] = dP_L[m] x[m,f]  (= dLoss/dY):
//     Q[c][j]    = sum_m Z_p[m,c] A[m][j]                               (the weight-gradient GEMM, 40 columns)
//     dZ_p[m,c]  = sum_j A[m][j] Ueff[c][j]                             (the data-gradient GEMM, reduction length 40)
//     dW_p = Q Bm,   dwsum_L[(n,f)] = sum_c Q[c][1+f] W_p[c][n] + ...,   dW_L[(n,f),:] = dwsum_L[(n,f)]
// Same function as the reference graph up to fp32 reassociation (tested against the fp64 oracle at every shape of the
// suite and at the benchmark shape, and against the general kernels: fil_cin mode 1); x^p and G^p never exist; all three
// big GEMMs of layer p shrink from H_p to F+1 (padded to 48 here) columns.  bench.py reports executed flops next to the
// algorithmic ones.
//
// MFMA shapes: the row-parallel forward and the weight-gradient kernel use v_mfma_f32_16x16x4_f32 (16-column blocks: 40 -> 48,
// a 32x32 tile would pad to 64); the data-gradient kernel keeps the 32x32x2 slot machinery of cin_dz3_kernel -- there the 40
// is the REDUCTION length and needs no padding at all.
#pragma once
#include "cin_kernels.h"

namespace fil {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over each row of 16 lanes (DPP only), valid in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);  // row_mirror
  return v;
}

// N consecutive dwords per lane through a raw buffer load of exactly that width.  The width matters: a 16-byte load of which
// only three components are used leaves a dead register in the middle of the operand queue, which the compiler (a) reuses as
// scratch -- every step then waits for the refill it has just issued before it may overwrite that register -- or (b) splits the
// vector into scalars that no longer sit in consecutive registers, so each refill is followed by a wait and copies.
template <int N>
struct DwordVec;
template <>
struct DwordVec<1> {
  typedef float T;
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
  }
  static __device__ __forceinline__ float get(const T& v, int) { return v; }
};
template <>
struct DwordVec<2> {
  typedef float T __attribute__((ext_vector_type(2)));
  typedef unsigned int U __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
  }
  static __device__ __forceinline__ float get(const T& v, int e) { return v[e]; }
};
template <>
struct DwordVec<3> {
  typedef float T __attribute__((ext_vector_type(3)));
  typedef unsigned int U __attribute__((ext_vector_type(3)));
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b96(r, vo, so, 0));
  }
  static __device__ __forceinline__ float get(const T& v, int e) { return v[e]; }
};
template <>
struct DwordVec<4> {
  typedef float T __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
  }
  static __device__ __forceinline__ float get(const T& v, int e) { return v[e]; }
};

// waits HERE for a pending load of v (see settle() in cin_kernels.h), any register width
template <typename T>
__device__ __forceinline__ void settle_any(T& v) {
  asm volatile("" : "+v"(v));
}

// ---------------------------------------------------------------------------------------------------------------------
// Ueff and its operand layouts.  One thread per (channel row c, column j <= F); 4 rows per workgroup, Bm staged in LDS.
//   Uf  (forward, B operand of 16x16x4):  [step s = h*JT4 + f/4][lane = (f&3)<<4 | (j&15)][component j>>4 < NCB]
//   Uz  (dZ, A operand of 32x32x2):       [tile][slot row i][wave half][JHp]: column j = half*JHp + s; slot row as in cin_pack_wz_kernel
//   consts: [0, JP) beff[j] = sum_n bias_p[n] Bm[j][n];  [JP] = sum_n' bias_L[n']
// Padding (f >= F, j > F, slots past the last channel) must be zero: the caller clears the buffers first.
static __global__ __launch_bounds__(256) void cin_tail_ueff_kernel(const float* __restrict__ Wp, const float* __restrict__ biasp,
                                                            const float* __restrict__ wsumL, const float* __restrict__ biasL, int HL,
                                                            float* __restrict__ Uf, float* __restrict__ Uz, float* __restrict__ consts,
                                                            int Hpp, int F, int Hq, int JT4, int JP, int JT, int JHp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // Bm [F+1][Hq+1]
  const int NCB = JP >> 4;
  const int J1 = F + 1, ld = Hq + 1;
  for (int idx = threadIdx.x; idx < J1 * Hq; idx += 256) {
    const int j = idx / Hq, n = idx - j * Hq;
    smem[j * ld + n] = j == 0 ? 1.f : wsumL[n * F + (j - 1)];
  }
  __syncthreads();
  const int Cp = Hpp * F;
  const int cl = threadIdx.x >> 6, j = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + cl;  // row Cp = the bias row (-> consts)
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float t = 0.f;
    for (int n = threadIdx.x; n < HL; n += 64) t += biasL[n];
    t = wave_sum(t);
    if (threadIdx.x == 0) consts[JP] = t;
  }
  if (c > Cp || j >= J1) return;
  const float* wrow = c < Cp ? Wp + (long)c * Hq : biasp;
  const float* brow = smem + j * ld;
  float t0 = 0.f, t1 = 0.f;
  int n = 0;
  for (; n + 1 < Hq; n += 2) {
    t0 = fmaf(wrow[n], brow[n], t0);
    t1 = fmaf(wrow[n + 1], brow[n + 1], t1);
  }
  if (n < Hq) t0 = fmaf(wrow[n], brow[n], t0);
  const float t = t0 + t1;
  if (c == Cp) {
    consts[j] = t;
    return;
  }
  const int h = c / F, f = c - h * F;
  Uf[((long)(h * JT4 + (f >> 2)) * 64 + (((f & 3) << 4) | (j & 15))) * NCB + (j >> 4)] = t;
  const int slot = h * JT + (f >> 1), hf = f & 1;
  const int tile = slot >> 4, rr = slot & 15;
  const int i = (rr & 3) + 8 * (rr >> 2) + 4 * hf;
  const int half = j >= JHp ? 1 : 0, s = j - half * JHp;
  Uz[((long)(tile * 32 + i) * 2 + half) * JHp + s] = t;
}

// queue depth of the forward's B-operand stream: a divisor of the steps per h (slot j4 % DEPTH must mean the same step in every h)
constexpr int tail_depth(int JT4) {
  return JT4 <= 10 ? JT4 : (JT4 == 12 ? 6 : (JT4 == 14 ? 7 : (JT4 == 15 ? 5 : 8)));
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward.  Y[m][j] = sum_{h,f} x^{p-1}[m,h] x[m,f] Ueff[(h,f)][j] + beff[j]  for j < 16*NCB, and the two sum-pools.
// Wave = RB blocks of 16 rows x NCB blocks of 16 columns.  Lane (i = lane&15, kq = lane>>4): A operand = row 16rb+i, reduction
// element kq of the step <-> field f = 4*j4 + kq;  B operand = Uf (one 16-byte load per step and lane: its NCB columns 16cb+i).
// The x fragment x[m, 4*j4 + kq] lives in registers (JT4 per row block), x^{p-1}[m,h] is one dword per h and row block.
template <int RB, int JT4, int NCB>
__global__ __launch_bounds__(256, 1) void cin_tail_fwd_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                              const float* __restrict__ Uf, const float* __restrict__ consts,
                                                              float* __restrict__ Y, int JP, float* __restrict__ pool_p,
                                                              float* __restrict__ pool_L, int M, int F, int Hp) {
  constexpr int DEPTH = tail_depth(JT4);
  static_assert(JT4 % DEPTH == 0, "queue depth must divide the steps per h");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int wrow0 = (blockIdx.x * 4 + wave) * (16 * RB);
  if (wrow0 >= M) return;
  long mq[RB];
  float xr[RB][JT4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = wrow0 + 16 * rb + i;
    mq[rb] = m < M ? m : M - 1;   // rows past M are clamped: computed, never stored
#pragma unroll
    for (int j4 = 0; j4 < JT4; ++j4) {
      const int f = 4 * j4 + kq;
      const float v = xT[mq[rb] * F + min(f, F - 1)];
      xr[rb][j4] = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & (f < F ? -1 : 0));
    }
  }
  f32x4 acc[RB][NCB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[rb][cb][e] = 0.f;
  typedef DwordVec<NCB> BV;
  constexpr int kStepBytes = 64 * NCB * 4;
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(Uf, (long)Hp * JT4 * kStepBytes);
  const int uo = lane * NCB * 4;
  typename BV::T q[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) q[d] = BV::load(ru, uo, d * kStepBytes);
  const float* xprow[RB];
  float xpv[RB], xpn[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    xprow[rb] = xpT + mq[rb] * xps;
    xpv[rb] = xprow[rb][0];
    xpn[rb] = 0.f;
  }
  // every prologue load is waited for once, HERE: the wait count at the top of the h loop then comes from the back edge alone
  // (the 10 refills of the previous h stay in flight); merged with the prologue's unordered loads it collapses to "drain the
  // queue at the top of every h" (see cin_fwd3_kernel)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) settle_any(q[d]);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) settle(xpv[rb]);
#pragma unroll 1
  for (int h = 0; h < Hp; ++h) {
    const bool more = h + 1 < Hp;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xpn[rb] = xprow[rb][more ? h + 1 : h];
    const int sh = h * JT4 * kStepBytes;                            // (scalar byte offsets: no per-load address arithmetic)
    const int shn = (more ? h + 1 : h) * JT4 * kStepBytes;
#pragma unroll
    for (int j4 = 0; j4 < JT4; ++j4) {
      const typename BV::T w = q[j4 % DEPTH];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float a = xpv[rb] * xr[rb][j4];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma16(a, BV::get(w, cb), acc[rb][cb]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int jn = j4 + DEPTH;   // refill AFTER the step's MFMAs: the load may land in the registers it replaces
      q[j4 % DEPTH] = jn < JT4 ? BV::load(ru, uo, sh + jn * kStepBytes) : BV::load(ru, uo, shn + (jn - JT4) * kStepBytes);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xpv[rb] = xpn[rb];
  }
  // ---- epilogue: + beff, store Y [M][JP] (16 consecutive columns per row and store), the two pools
  float be[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) be[cb] = consts[16 * cb + i];
  const float bsumL = consts[JP];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = wrow0 + 16 * rb + 4 * kq + reg;
      const long mc = m < M ? m : M - 1;
      float e = 0.f;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int j = 16 * cb + i;
        const float y = acc[rb][cb][reg] + be[cb];
        if (m < M) Y[(long)m * JP + j] = y;
        const float xv = xT[mc * F + min(max(j - 1, 0), F - 1)];
        e = fmaf(y, __builtin_bit_cast(float, __builtin_bit_cast(int, xv) & ((j >= 1 && j <= F) ? -1 : 0)), e);
      }
      e = row16_sum(e);
      if (i == 0 && m < M) {
        pool_p[m] = acc[rb][0][reg] + be[0];
        pool_L[m] = e + bsumL;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// A = dLoss/dY as the weight-gradient kernel's B operand: Apk[m][jj < 16][cb < NCB] = A[m][16cb + jj]
//   A[m][0] = dP_p[m];  A[m][1+f] = dP_L[m] x[m,f];  A[m][F+1] = dP_L[m] (its column sum is dbias_L);  zero beyond.
static __global__ __launch_bounds__(256) void cin_tail_a_kernel(const float* __restrict__ xT, const float* __restrict__ dP, int ldp, int K,
                                                         int lp, int lL, float* __restrict__ Apk, int M, int F, int NCB) {
  const long total = (long)M * 16 * NCB;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cb = (int)(idx % NCB);
    const long t = idx / NCB;
    const long m = t >> 4;
    const int j = 16 * cb + (int)(t & 15);
    const long b = m / K;
    const int k = (int)(m - b * K);
    const float dpl = dP[b * ldp + lL * K + k];
    Apk[idx] = j == 0 ? dP[b * ldp + lp * K + k] : (j <= F ? dpl * xT[m * F + (j - 1)] : (j == F + 1 ? dpl : 0.f));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM of the fused tail: Q[c][j] = sum_m Z_p[m,c] A[m][j]  (c <= C_p: row C_p is an all-ones channel, whose
// sums are dbeff / dbias_L), one row split per workgroup item, partials reduced in fixed order by cin_reduce_kernel.
// Wave = CBW blocks of 16 channel rows x NCB blocks of 16 columns; step = 4 rows of m (reduction element kq = lane>>4).
//   A operand: x^{p-1}[m,h_c] * x[m,f_c]  (two dword gathers per channel block, raw buffer loads with scalar row offsets)
//   B operand: Apk[m][i][0..NCB)          (one load of exactly NCB dwords)
constexpr int kTailDwDepth = 8;
constexpr int kTailCbw = 4;

template <int NCB, bool SETTLE = false, int CBW = kTailCbw, int DEPTH = kTailDwDepth>
__global__ __launch_bounds__(256, 2) void cin_tail_dw_kernel(const float* __restrict__ Apk, const float* __restrict__ xT,
                                                              const float* __restrict__ xpT, int xps, float* __restrict__ part, int M,
                                                              int F, int Hp, int JP, int rows_per_split, int blocks_x, int items) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int Cp = Hp * F, C1 = Cp + 1;
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, as in cin_dw3_kernel
  if (item >= items) return;
  const int bx = item % blocks_x;
  const int split = item / blocks_x;
  const int c0 = (bx * 4 + wave) * (16 * CBW);
  if (c0 >= C1) return;
  const int m_lo = split * rows_per_split;
  const int m_hi = min(M, m_lo + rows_per_split);
  const long mrem = (long)M - m_lo;
  typedef DwordVec<NCB> BV;
  constexpr int kRowBytes = 16 * NCB * 4;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Apk + (long)m_lo * 16 * NCB, mrem * kRowBytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xT + (long)m_lo * F, mrem * F * 4);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(xpT + (long)m_lo * xps, mrem * xps * 4);
  int fo[CBW], ho[CBW];
  bool cv[CBW], ones[CBW];
#pragma unroll
  for (int cbk = 0; cbk < CBW; ++cbk) {
    const int c = c0 + 16 * cbk + i;
    cv[cbk] = c < C1;
    ones[cbk] = c == Cp;
    const int cc = min(c, Cp - 1);
    const int hh = cc / F, ff = cc - hh * F;
    ho[cbk] = (kq * xps + hh) * 4;   // byte offsets of the lane's column inside row (m_lo + kq)
    fo[cbk] = (kq * F + ff) * 4;
  }
  const int ao = (kq * 16 + i) * NCB * 4;
  const int steps = (m_hi - m_lo + 3) >> 2;
  const int groups = (steps + DEPTH - 1) / DEPTH;
  const int mlane = m_lo + kq;

  f32x4 acc[CBW][NCB];
#pragma unroll
  for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[cbk][cb][e] = 0.f;

  typename BV::T qa[DEPTH];
  float qx[DEPTH][CBW], qp[DEPTH][CBW];
  auto fetch = [&](int s, typename BV::T& a4, float (&xv)[CBW], float (&pv)[CBW]) {
    const int row = 4 * s;  // uniform, relative to the split's first row
    a4 = BV::load(ra, ao, row * kRowBytes);
#pragma unroll
    for (int cbk = 0; cbk < CBW; ++cbk) {
      xv[cbk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, fo[cbk], row * F * 4, 0));
      pv[cbk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, ho[cbk], row * xps * 4, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, qa[d], qx[d], qp[d]);
  if constexpr (SETTLE) {   // (wait for the prologue's loads once, before the loop: see cin_tail_fwd_kernel)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      settle_any(qa[d]);
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk) {
        settle(qx[d][cbk]);
        settle(qp[d][cbk]);
      }
    }
  }
#pragma unroll 1
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int s = g * DEPTH + d;
      const typename BV::T a4 = qa[d];
      const bool live = mlane + 4 * s < m_hi;
      float a[CBW];
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk) {
        const float z = ones[cbk] ? 1.f : qx[d][cbk] * qp[d][cbk];
        a[cbk] = (live && cv[cbk]) ? z : 0.f;
      }
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cbk][cb] = mfma16(a[cbk], BV::get(a4, cb), acc[cbk][cb]);
      __builtin_amdgcn_sched_barrier(0);
      fetch(s + DEPTH, qa[d], qx[d], qp[d]);   // refill after the step's MFMAs (lands in place, see cin_dw3_kernel)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D[row = 4*kq + reg (channel of the block)][col = i]
  float* pout = part + (long)split * C1 * JP;
#pragma unroll
  for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int c = c0 + 16 * cbk + 4 * kq + reg;
      if (c < C1) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) pout[(long)c * JP + 16 * cb + i] = acc[cbk][cb][reg];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Parameter gradients from Q [C_p + 1][JP].  One workgroup per kTailPc channel rows:
//   dW_p[c][n]  = sum_{j <= F} Q[c][j] Bm[j][n]
//   partB[blk][(n,f)] = sum_{c in blk} Q[c][1+f] W_p[c][n]           (-> dwsum_L, reduced by cin_tail_fill_kernel)
//   workgroup 0: dbias_p[n] = sum_j Q[C_p][j] Bm[j][n]
constexpr int kTailPc = 32;
static __global__ __launch_bounds__(256) void cin_tail_params_kernel(const float* __restrict__ Q, const float* __restrict__ Wp,
                                                              const float* __restrict__ wsumL, float* __restrict__ dWp,
                                                              float* __restrict__ dbiasp, float* __restrict__ partB, int Cp, int F, int Hq,
                                                              int JP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // Bm [F+1][Hq+1] | Q rows [kTailPc + 1][JP]
  const int J1 = F + 1, ld = Hq + 1;
  float* bm = smem;
  float* qs = smem + J1 * ld;
  const int c0 = blockIdx.x * kTailPc;
  const int nc = min(kTailPc, Cp - c0);
  for (int idx = threadIdx.x; idx < J1 * Hq; idx += 256) {
    const int j = idx / Hq, n = idx - j * Hq;
    bm[j * ld + n] = j == 0 ? 1.f : wsumL[n * F + (j - 1)];
  }
  for (int idx = threadIdx.x; idx < (kTailPc + 1) * JP; idx += 256) {
    const int cl = idx / JP, j = idx - cl * JP;
    const int c = cl < kTailPc ? c0 + cl : Cp;   // last staged row: the ones channel
    qs[idx] = (cl == kTailPc || cl < nc) ? Q[(long)c * JP + j] : 0.f;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nc * Hq; idx += 256) {
    const int cl = idx / Hq, n = idx - cl * Hq;
    const float* qrow = qs + cl * JP;
    float t = 0.f;
    for (int j = 0; j < J1; ++j) t = fmaf(qrow[j], bm[j * ld + n], t);
    dWp[(long)(c0 + cl) * Hq + n] = t;
  }
  if (blockIdx.x == 0) {
    const float* qrow = qs + kTailPc * JP;
    for (int n = threadIdx.x; n < Hq; n += 256) {
      float t = 0.f;
      for (int j = 0; j < J1; ++j) t = fmaf(qrow[j], bm[j * ld + n], t);
      dbiasp[n] = t;
    }
  }
  float* pb = partB + (long)blockIdx.x * Hq * F;
  for (int idx = threadIdx.x; idx < Hq * F; idx += 256) {
    const int n = idx / F, f = idx - n * F;
    float t = 0.f;
    for (int cl = 0; cl < nc; ++cl) t = fmaf(qs[cl * JP + 1 + f], Wp[(long)(c0 + cl) * Hq + n], t);
    pb[idx] = t;
  }
}

// dwsum_L[(n,f)] = sum_blk partB[blk][(n,f)] + Q[C_p][1+f] bias_p[n];  dW_L[(n,f)][n'] = dwsum_L[(n,f)] for every n';
// workgroup 0: dbias_L[n'] = Q[C_p][F+1] (= sum_m dP_L[m]).  64 rows per workgroup, partial sums folded through LDS in fixed order.
static __global__ __launch_bounds__(256) void cin_tail_fill_kernel(const float* __restrict__ partB, int parts, const float* __restrict__ Q,
                                                            const float* __restrict__ biasp, float* __restrict__ dWL,
                                                            float* __restrict__ dbiasL, int Cp, int F, int Hq, int HL, int JP) {
  __shared__ float red[4][64];
  __shared__ float val[64];
  const int rows = Hq * F;
  const int r0 = blockIdx.x * 64;
  const int rl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int row = r0 + rl;
  float t0 = 0.f, t1 = 0.f;
  if (row < rows) {
    int p = g;
    for (; p + 4 < parts; p += 8) {
      t0 += partB[(long)p * rows + row];
      t1 += partB[(long)(p + 4) * rows + row];
    }
    if (p < parts) t0 += partB[(long)p * rows + row];
  }
  red[g][rl] = t0 + t1;
  __syncthreads();
  if (g == 0 && row < rows) {
    const int n = row / F, f = row - n * F;
    val[rl] = ((red[0][rl] + red[1][rl]) + (red[2][rl] + red[3][rl])) + Q[(long)Cp * JP + 1 + f] * biasp[n];
  }
  __syncthreads();
  const int nrow = min(64, rows - r0);
  for (int idx = threadIdx.x; idx < nrow * HL; idx += 256) {
    const int r = idx / HL;
    dWL[(long)r0 * HL + idx] = val[r];
  }
  if (blockIdx.x == 0) {
    const float s = Q[(long)Cp * JP + F + 1];
    for (int n = threadIdx.x; n < HL; n += 256) dbiasL[n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Data-gradient GEMM of the fused tail, the slot machinery of cin_dz3_kernel with the 40-long reduction:
//   dZ^T tile (32 slot rows x 32 rows m) = Uz tile (A operand, streamed: NQ float4 per tile and lane) x A^T (B operand: the
//   lane's own row of A, generated in registers: JHp = 4*NQ values per wave half)
//   G^{p-1}[m,h] = sum_f dZ[(h,f),m] x[m,f] + dP_{p-1}[m];   dX[m,f] = dP_L[m] Y[m][1+f] + sum_h dZ[(h,f),m] x^{p-1}[m,h]
// With 4*NQ MFMA steps per tile instead of 64 the register contraction (2 FMAs per dZ element) is no longer a side show: the x
// fragment and the dX accumulators live in REGISTERS here (the lane's A row is 20 registers instead of 64), so a slot costs its
// two FMAs and nothing else; 32 rows per wave, two waves per SIMD.
template <int JT, int NQ>
__global__ __launch_bounds__(256, 2) void cin_tail_dz_kernel(const float* __restrict__ Uz, const float* __restrict__ xT,
                                                              const float* __restrict__ xpT, int xps, const float* __restrict__ Y, int JP,
                                                              const float* __restrict__ dP, int ldp, int K, int lp, int lL,
                                                              float* __restrict__ GprevT, int HSp, float* __restrict__ dxT, int M, int F,
                                                              int Hp, int periods) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // dX staging [JT][256] | per-wave line buffers [32][kGlStride]
  constexpr int JHp = 4 * NQ;
  constexpr int P = JT / gcd_c(16, JT);
  constexpr int HPP = 16 * P / JT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M) return;
  float* gl = smem + JT * 256 + wave * (32 * kGlStride);
  const int m = wrow0 + r;
  const bool vq = m < M;
  const long mq = vq ? m : M - 1;
  const int keepm = vq ? -1 : 0;
  const long bb = mq / K;
  const int kk = (int)(mq - bb * K);
  const float dpl = __builtin_bit_cast(float, __builtin_bit_cast(int, dP[bb * ldp + lL * K + kk]) & keepm);
  const float dpp = __builtin_bit_cast(float, __builtin_bit_cast(int, dP[bb * ldp + lp * K + kk]) & keepm);
  const float dpprev = __builtin_bit_cast(float, __builtin_bit_cast(int, dP[bb * ldp + (lp - 1) * K + kk]) & keepm);
  float xr[JT], dxa[JT], areg[JHp];
  {
    float xt[JT], yt[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int f = min(2 * j + half, F - 1);
      xt[j] = xT[mq * F + f];
      yt[j] = Y[mq * JP + 1 + f];
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int keep = (vq && 2 * j + half < F) ? -1 : 0;
      xr[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, xt[j]) & keep);
      dxa[j] = dpl * __builtin_bit_cast(float, __builtin_bit_cast(int, yt[j]) & keep);
    }
    float at[JHp];
#pragma unroll
    for (int s = 0; s < JHp; ++s) {
      const int j = half * JHp + s;
      at[s] = xT[mq * F + min(max(j - 1, 0), F - 1)];
    }
#pragma unroll
    for (int s = 0; s < JHp; ++s) {
      const int j = half * JHp + s;
      const float v = dpl * __builtin_bit_cast(float, __builtin_bit_cast(int, at[s]) & ((j >= 1 && j <= F) ? -1 : 0));
      areg[s] = j == 0 ? dpp : v;
    }
  }
  const float4* wz = reinterpret_cast<const float4*>(Uz) + ((long)r * 2 + half) * NQ;
  constexpr long kTileStride = 64L * NQ;  // float4 per tile
  float4 q[NQ];
#pragma unroll
  for (int s4 = 0; s4 < NQ; ++s4) q[s4] = wz[s4];
  float gx = 0.f;
  f32x16 dprev;
#pragma unroll
  for (int e = 0; e < 16; ++e) dprev[e] = 0.f;
  float xprev[HPP], xcur[HPP], gout[HPP];
#pragma unroll
  for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl] = gout[hl] = 0.f;
  int hprev = -HPP;

  // contraction of slot rr of tile tpp (period base hb): compile-time (hl, j) after unrolling
  auto slot_apply = [&](const f32x16& d, const float (&xpv)[HPP], int hb, int tpp, int rr) {
    const int sp = 16 * tpp + rr;
    const int hl = sp / JT, j = sp % JT;
    const float dz = d[rr];
    gx = fmaf(dz, xr[j], gx);
    dxa[j] = fmaf(dz, xpv[hl], dxa[j]);
    if (j == JT - 1) {
      gout[hl] = lane_halves_sum(gx) + dpprev;
      gx = 0.f;
      if (hl == HPP - 1 && half == 0 && hb >= 0) {
        float* bl = gl + r * kGlStride + (hb & 31);
        if constexpr (HPP == 4) *reinterpret_cast<float4*>(bl) = make_float4(gout[0], gout[1], gout[2], gout[3]);
        else if constexpr (HPP == 2) *reinterpret_cast<float2*>(bl) = make_float2(gout[0], gout[1]);
        else bl[0] = gout[0];
      }
    }
  };
  auto flush_line = [&](int line) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane, row = idx >> 3, c4 = idx & 7;
      const int mm = wrow0 + row, col = 32 * line + 4 * c4;
      const float4 v = *reinterpret_cast<const float4*>(gl + row * kGlStride + 4 * c4);
      if (mm < M && col < Hp) {
        float* dst = GprevT + (long)mm * HSp + col;
        if (col + 3 < Hp) *reinterpret_cast<float4*>(dst) = v;
        else {
          dst[0] = v.x;
          if (col + 1 < Hp) dst[1] = v.y;
          if (col + 2 < Hp) dst[2] = v.z;
        }
      }
    }
  };

#pragma unroll 1
  for (int per = 0; per < periods; ++per) {
    const int hbase = per * HPP;
#pragma unroll
    for (int hl = 0; hl < HPP; ++hl) {
      const float xv = xpT[mq * xps + min(hbase + hl, Hp - 1)];
      const int keep = (vq && hbase + hl < Hp) ? -1 : 0;
      xcur[hl] = __builtin_bit_cast(float, __builtin_bit_cast(int, xv) & keep);
    }
#pragma unroll
    for (int tp = 0; tp < P; ++tp) {
      const float4* wnext = wz + ((long)per * P + tp + 1) * kTileStride;   // (the stream is allocated one tile past the end)
      f32x16 d;
#pragma unroll
      for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < NQ; ++s4) {
        const float4 w = q[s4];
        q[s4] = wnext[s4];
        d = mfma32(w.x, areg[4 * s4 + 0], d);
        d = mfma32(w.y, areg[4 * s4 + 1], d);
        d = mfma32(w.z, areg[4 * s4 + 2], d);
        d = mfma32(w.w, areg[4 * s4 + 3], d);
        // the previous tile's 16 slots, spread over this tile's NQ step groups
#pragma unroll
        for (int sl = (16 * s4) / NQ; sl < (16 * (s4 + 1)) / NQ; ++sl) {
          if (tp == 0) slot_apply(dprev, xprev, hprev, P - 1, sl);
          else slot_apply(dprev, xcur, hbase, tp - 1, sl);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      dprev = d;
      if (tp == 0 && hprev >= 0 && ((hprev + HPP) & 31) == 0) flush_line(hprev >> 5);
    }
#pragma unroll
    for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl];
    hprev = hbase;
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) slot_apply(dprev, xprev, hprev, P - 1, rr);
  if (hprev >= 0) flush_line(hprev >> 5);   // the last (possibly partial) line
  // dX rows of the wave are contiguous in dxT ([32 rows][F]): staged in LDS, stored cooperatively as whole lines
  float* dxs = smem + tid;
#pragma unroll
  for (int j = 0; j < JT; ++j) dxs[j * 256] = dxa[j];
  __builtin_amdgcn_wave_barrier();
  const float* dsc = smem + wave * 64;
  const int nrow = min(32, M - wrow0);
  float* dst = dxT + (long)wrow0 * F;
  for (int idx = lane; idx < nrow * F; idx += 64) {
    const int rr = idx / F, f = idx - rr * F;
    dst[idx] = dsc[(f >> 1) * 256 + (f & 1) * 32 + rr];
  }
}

}  // namespace fil
