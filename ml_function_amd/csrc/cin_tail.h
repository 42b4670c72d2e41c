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

#include <type_traits>

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
// bmT[f][n] = wsum_L[(n,f)] = sum_n' W_L[(n*F + f), n']: the pooled weights of the last layer, stored so that Bm[1+f][.] is a
// contiguous row (the staging loops below are coalesced).  One half wave per row of W_L.
__device__ __forceinline__ void cin_tail_wsum_body(const float* __restrict__ WL, float* __restrict__ bmT, int Hq, int F, int HL, int bid) {
  const int row = bid * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  const int C = Hq * F;
  float t = 0.f;
  if (row < C)
    for (int n = l; n < HL; n += 32) t += WL[(long)row * HL + n];
  t = half_wave_sum(t);
  if (row < C && l == 0) {
    const int n = row / F, f = row - n * F;
    bmT[f * Hq + n] = t;
  }
}
static __global__ __launch_bounds__(256) void cin_tail_wsum_kernel(const float* __restrict__ WL, float* __restrict__ bmT, int Hq, int F, int HL) {
  cin_tail_wsum_body(WL, bmT, Hq, F, HL, blockIdx.x);
}

// One launch for the forward's independent preparation jobs (each ~4-9 us as a launch of its own, none of them busy for more
// than a fraction of that): [0, nt) x -> xT transposes (one sample per workgroup); [nt, nt+npk) the pair-symmetric first
// layer's packed weights; then the pooled weights of the last layer (bmT, 8 rows per workgroup); the rest clears the fused
// tail's operand buffers.
static __global__ __launch_bounds__(256) void cin_fwd_prep_kernel(const float* __restrict__ x, float* __restrict__ xT, int F, int K, int nt,
                                                           const float* __restrict__ W0, float* __restrict__ Wf, int H0, int JT2,
                                                           int chunks, int npk, const float* __restrict__ WL, float* __restrict__ bmT,
                                                           int Hq, int HL, int nws, float4* __restrict__ zero, long nzero4,
                                                           float* __restrict__ x2T, int XL, int xt_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < nt) {
    if (xt_in) cin_wrap_rows_body(x, x2T, F, K, b, XL);
    else cin_transpose_in_body(x, xT, F, K, b, smem, x2T, XL);
  } else if (b < nt + npk) {
    cin_pack_wf_sym_body(W0, Wf, F, H0, JT2, chunks, b - nt, npk);
  } else if (b < nt + npk + nws) {
    cin_tail_wsum_body(WL, bmT, Hq, F, HL, b - nt - npk);
  } else {
    const int nz = gridDim.x - (nt + npk + nws);
    for (long i = (long)(b - nt - npk - nws) * 256 + threadIdx.x; i < nzero4; i += (long)nz * 256) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// stages Bm [F+1][ld] (row 0 = ones, row 1+f = bmT[f]) into LDS
__device__ __forceinline__ void tail_stage_bm(const float* __restrict__ bmT, float* bm, int F, int Hq, int ld) {
  for (int idx = threadIdx.x; idx < (F + 1) * Hq; idx += blockDim.x) {
    const int j = idx / Hq, n = idx - j * Hq;
    bm[j * ld + n] = j == 0 ? 1.f : bmT[idx - Hq];
  }
}

// Ueff and its operand layouts.  kTailUc channel rows per workgroup; thread (row cl = tid/16, column group jg = tid%16) computes
// the columns j = jg, jg+16, ... of its row, so a W_p row is read once per 16 lanes (broadcast) and Bm comes from LDS.
//   Uf  (forward, B operand of 16x16x4):  [step s = h*JT4 + f/4][lane = (f&3)<<4 | (j&15)][component j>>4 < NCB]
//   Uz  (dZ, A operand of 32x32x2):       [tile][slot row i][wave half][JHp]: column j = half*JHp + s; slot row as in cin_pack_wz_kernel
//   consts: [0, JP) beff[j] = sum_n bias_p[n] Bm[j][n];  [JP] = sum_n' bias_L[n']
// Padding (f >= F, j > F, slots past the last channel) must be zero: the caller clears the buffers first.
constexpr int kTailUc = 16;
static __global__ __launch_bounds__(256) void cin_tail_ueff_kernel(const float* __restrict__ Wp, const float* __restrict__ biasp,
                                                            const float* __restrict__ bmT, const float* __restrict__ biasL, int HL,
                                                            float* __restrict__ Uf, float* __restrict__ Uz, float* __restrict__ consts,
                                                            int Hpp, int F, int Hq, int JT4, int JP, int JT, int JHp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // Bm [F+1][ld], ld = Hq rounded up to 4, + 4: 16-byte aligned rows
  const int NCB = JP >> 4;
  const int J1 = F + 1, ld = ((Hq + 3) & ~3) + 4;
  for (int idx = threadIdx.x; idx < J1 * ld; idx += 256) {
    const int j = idx / ld, n = idx - j * ld;
    smem[idx] = n < Hq ? (j == 0 ? 1.f : bmT[(j - 1) * Hq + n]) : 0.f;   // (zero pad: the 4-wide loop below runs over it)
  }
  __syncthreads();
  const int Cp = Hpp * F;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float t = 0.f;
    for (int n = threadIdx.x; n < HL; n += 64) t += biasL[n];
    t = wave_sum(t);
    if (threadIdx.x == 0) consts[JP] = t;
  }
  const int cl = threadIdx.x >> 4, jg = threadIdx.x & 15;
  const int c = blockIdx.x * kTailUc + cl;  // row Cp = the bias row (-> consts)
  if (c > Cp) return;
  const float* wrow = c < Cp ? Wp + (long)c * Hq : biasp;
  const bool vec = (Hq & 3) == 0;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  const float* brow[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) brow[cb] = smem + min(16 * cb + jg, F) * ld;
  for (int n = 0; n < Hq; n += 4) {
    float w[4];
    if (vec) {
      const float4 w4 = *reinterpret_cast<const float4*>(wrow + n);
      w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = n + e < Hq ? wrow[n + e] : 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      if (cb < NCB) {
        const float4 b4 = *reinterpret_cast<const float4*>(brow[cb] + n);
        t[cb] = fmaf(w[0], b4.x, t[cb]);
        t[cb] = fmaf(w[1], b4.y, t[cb]);
        t[cb] = fmaf(w[2], b4.z, t[cb]);
        t[cb] = fmaf(w[3], b4.w, t[cb]);
      }
    }
  }
  const int h = c / F, f = c - h * F;
  const int slot = h * JT + (f >> 1), hf = f & 1;
  const int tile = slot >> 4, rr = slot & 15;
  const int i = (rr & 3) + 8 * (rr >> 2) + 4 * hf;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int j = 16 * cb + jg;
    if (cb >= NCB || j >= J1) continue;
    if (c == Cp) {
      consts[j] = t[cb];
      continue;
    }
    Uf[((long)(h * JT4 + (f >> 2)) * 64 + (((f & 3) << 4) | jg)) * NCB + cb] = t[cb];
    const int half = j >= JHp ? 1 : 0, s2 = j - half * JHp;
    Uz[((long)(tile * 32 + i) * 2 + half) * JHp + s2] = t[cb];
  }
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
//
// KS > 1 (small M: fewer row blocks than SIMDs): KS waves of the workgroup share one row block and split the reduction over h
// between them (wave kpart takes h in [kpart*ceil(Hp/KS), ...)); the partial accumulators are folded through LDS in wave order
// and the first wave of the group runs the epilogue.  A wave reduces over ALL channels of its rows whatever M is, so without the
// split the kernel's time stops shrinking below one row block per SIMD (DESIGN.md section 6, strong scaling).
template <int RB, int JT4, int NCB, int KS = 1>
__global__ __launch_bounds__(256, 1) void cin_tail_fwd_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                              const float* __restrict__ Uf, const float* __restrict__ consts,
                                                              float* __restrict__ Y, int JP, float* __restrict__ pool_p,
                                                              float* __restrict__ pool_L, int M, int F, int Hp) {
  constexpr int DEPTH = tail_depth(JT4);
  static_assert(JT4 % DEPTH == 0, "queue depth must divide the steps per h");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int kpart = wave % KS;
  const int wrow0 = ((blockIdx.x * 4 + wave) / KS) * (16 * RB);
  const int hq = (Hp + KS - 1) / KS;
  const int h_lo = KS == 1 ? 0 : min(Hp, kpart * hq);
  const int h_hi = KS == 1 ? Hp : (wrow0 < M ? min(Hp, h_lo + hq) : h_lo);   // (row blocks past M: no work, but the barrier below is met)
  if (KS == 1 && wrow0 >= M) return;
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
  const int hs = min(h_lo, Hp - 1);   // first h of this wave (clamped: an empty range still issues its in-range prologue loads)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) q[d] = BV::load(ru, uo, (hs * JT4 + d) * kStepBytes);
  const float* xprow[RB];
  float xpv[RB], xpn[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    xprow[rb] = xpT + mq[rb] * xps;
    xpv[rb] = xprow[rb][hs];
    xpn[rb] = 0.f;
  }
  // every prologue load is waited for once, HERE: the wait count at the top of the h loop then comes from the back edge alone
  // (the 10 refills of the previous h stay in flight); merged with the prologue's unordered loads it collapses to "drain the
  // queue at the top of every h" (see cin_fwd3_kernel)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) settle_any(q[d]);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) settle(xpv[rb]);
  // The A operands of step s+1 are computed as ONE block of RB multiplies in front of step s's MFMAs and consumed a step later:
  // a v_mul whose result feeds the very next MFMA costs ~14 cycles of matrix time (VALU -> MFMA operand wait states; 128 instead
  // of 145 TFLOP/s in tools/probe_mfma16.hip at this 4 x 3 shape), a block of independent ones ~1.5 each (143).
  float an[RB], ac[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) ac[rb] = xpv[rb] * xr[rb][0];
#pragma unroll 1
  for (int h = h_lo; h < h_hi; ++h) {
    const bool more = h + 1 < Hp;   // (the prefetch may run into the next wave's h range: in range, never used)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xpn[rb] = xprow[rb][more ? h + 1 : h];
    const int sh = h * JT4 * kStepBytes;                            // (scalar byte offsets: no per-load address arithmetic)
    const int shn = (more ? h + 1 : h) * JT4 * kStepBytes;
#pragma unroll
    for (int j4 = 0; j4 < JT4; ++j4) {
      const typename BV::T w = q[j4 % DEPTH];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) an[rb] = (j4 + 1 < JT4 ? xpv[rb] : xpn[rb]) * xr[rb][(j4 + 1) % JT4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mfma16(ac[rb], BV::get(w, cb), acc[rb][cb]);
      __builtin_amdgcn_sched_barrier(0);
      const int jn = j4 + DEPTH;   // refill AFTER the step's MFMAs: the load may land in the registers it replaces
      q[j4 % DEPTH] = jn < JT4 ? BV::load(ru, uo, sh + jn * kStepBytes) : BV::load(ru, uo, shn + (jn - JT4) * kStepBytes);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) ac[rb] = an[rb];
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xpv[rb] = xpn[rb];
  }
  if constexpr (KS > 1) {
    __shared__ float fold[4][RB * NCB * 4][64];
    if (kpart > 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) fold[wave][(rb * NCB + cb) * 4 + e][lane] = acc[rb][cb][e];
    }
    __syncthreads();
    if (kpart > 0 || wrow0 >= M) return;
#pragma unroll
    for (int k = 1; k < KS; ++k)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[rb][cb][e] += fold[wave + k][(rb * NCB + cb) * 4 + e][lane];
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
__device__ __forceinline__ void cin_tail_a_body(const float* __restrict__ xT, const float* __restrict__ dP, int ldp, int K, int lp, int lL,
                                                float* __restrict__ Apk, int M, int F, int NCB, int bid, int nblocks) {
  const long total = (long)M * 16 * NCB;
  for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblocks * 256) {
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
// One launch for three independent jobs at the start of the backward: [0, na) A = dLoss/dY; [na, na+nh) the fixed-order sum of
// the dense head's gradient partials; the rest (if any) the slot-ordered pair weights of the first layer's dZ kernel.
static __global__ __launch_bounds__(256) void cin_tail_pre_kernel(const float* __restrict__ xT, const float* __restrict__ dP, int ldp, int K,
                                                           int lp, int lL, float* __restrict__ Apk, int M, int F, int NCB, int na,
                                                           const float* __restrict__ hpart, float* __restrict__ ddense_w,
                                                           float* __restrict__ ddense_b, int LK, int hparts, int nh,
                                                           const float* __restrict__ W0, float* __restrict__ Wz, int H0, int JTs, int HS0,
                                                           int tiles) {
  const int b = blockIdx.x;
  if (b < na) cin_tail_a_body(xT, dP, ldp, K, lp, lL, Apk, M, F, NCB, b, na);
  else if (b < na + nh) cin_reduce_body(hpart, ddense_w, (long)LK + 1, hparts, ddense_b, (long)LK, b - na);
  else cin_pack_wz_sym_body(W0, Wz, F, H0, JTs, HS0, tiles, b - na - nh, gridDim.x - na - nh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM of the fused tail: Q[c][j] = sum_m Z_p[m,c] A[m][j]  (c <= C_p: row C_p is an all-ones channel, whose
// sums are dbeff / dbias_L), one row split per workgroup item, partials reduced in fixed order by cin_reduce_kernel.
// Wave = CBW blocks of 16 channel rows x NCB blocks of 16 columns over a quarter of the workgroup's row split; step = 4 rows of m
// (reduction element kq = lane>>4).
//   A operand: x^{p-1}[m,h_c] * x[m,f_c]  (two dword gathers per channel block, raw buffer loads with scalar row offsets)
//   B operand: Apk[m][i][0..NCB)          (one load of exactly NCB dwords)
constexpr int kTailDwDepth = 8;
constexpr int kTailCbw = 4;

template <int NCB, bool SETTLE = false, int CBW = kTailCbw, int DEPTH = kTailDwDepth>
__global__ __launch_bounds__(256, 2) void cin_tail_dw_kernel(const float* __restrict__ Apk, const float* __restrict__ xT,
                                                              const float* __restrict__ xpT, int xps, float* __restrict__ part, int M,
                                                              int F, int Hp, int JP, int rows_per_split, int blocks_x, int items) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int Cp = Hp * F, C1 = Cp + 1;
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware, as in cin_dw3_kernel
  if (item >= items) return;
  // Workgroup = one block of 16*CBW channel rows x one row split; its four waves take the four quarters of the split and their
  // accumulators are folded through LDS (fixed order) into ONE partial per workgroup: a quarter of the partial sums to write and
  // to reduce afterwards for the same number of waves in flight.
  const int bx = item % blocks_x;
  const int split = item / blocks_x;
  const int c0 = bx * (16 * CBW);
  const int quarter = rows_per_split >> 2;             // (a multiple of 4 * DEPTH rows: the host rounds the split to that)
  const int m_lo = min(M, split * rows_per_split + wave * quarter);
  const int m_hi = min(M, min(m_lo + quarter, (split + 1) * rows_per_split));
  // the three descriptors cover exactly the split's rows [m_lo, m_hi): anything past them -- the last step's spare rows, the
  // prefetch beyond the end -- reads as zero, so the loop needs no row masks at all (0 * 0 products; 1 * A = 0 in the ones row)
  const long mrem = (long)m_hi - m_lo;
  typedef DwordVec<NCB> BV;
  constexpr int kRowBytes = 16 * NCB * 4;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Apk + (long)m_lo * 16 * NCB, mrem * kRowBytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xT + (long)m_lo * F, mrem * F * 4);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(xpT + (long)m_lo * xps, mrem * xps * 4);
  int fo[CBW], ho[CBW];
  bool ones[CBW];
#pragma unroll
  for (int cbk = 0; cbk < CBW; ++cbk) {
    const int c = c0 + 16 * cbk + i;
    ones[cbk] = c == Cp;
    const int cc = min(c, Cp - 1);   // (channels past the end compute finite values that are never stored)
    const int hh = cc / F, ff = cc - hh * F;
    ho[cbk] = (kq * xps + hh) * 4;   // byte offsets of the lane's column inside row (m_lo + kq)
    fo[cbk] = (kq * F + ff) * 4;
  }
  const int ao = (kq * 16 + i) * NCB * 4;
  const int steps = (m_hi - m_lo + 3) >> 2;
  const int groups = (steps + DEPTH - 1) / DEPTH;

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
  // Main loop.  The generated operands of step s+1 are computed as one block in front of step s's MFMAs (see cin_tail_fwd_kernel);
  // only the wave that owns the all-ones channel row C_p pays for the select (a wave-uniform choice of the loop body).
  auto run = [&](auto ones_tag) {
    constexpr bool ONES = decltype(ones_tag)::value;
    auto make_a = [&](int d, float (&a)[CBW]) {
#pragma unroll
      for (int cbk = 0; cbk < CBW; ++cbk) {
        const float z = qx[d][cbk] * qp[d][cbk];
        a[cbk] = (ONES && ones[cbk]) ? 1.f : z;
      }
    };
    float ac[CBW], an[CBW];
    make_a(0, ac);
#pragma unroll 1
    for (int g = 0; g < groups; ++g) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int s = g * DEPTH + d;
        const typename BV::T a4 = qa[d];
        make_a((d + 1) % DEPTH, an);   // (slot 0 was refilled with the next group's first step at d == 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cbk][cb] = mfma16(ac[cbk], BV::get(a4, cb), acc[cbk][cb]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(s + DEPTH, qa[d], qx[d], qp[d]);   // refill after the step's MFMAs (lands in place, see cin_dw3_kernel)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cbk = 0; cbk < CBW; ++cbk) ac[cbk] = an[cbk];
      }
    }
  };
  if (c0 + 16 * CBW > Cp) run(std::true_type{});
  else run(std::false_type{});
  // fold the four waves' sums: waves 1..3 park theirs in LDS, wave 0 adds them in wave order and stores
  __shared__ float fold[3][CBW * NCB * 4][64];
  if (wave > 0) {
#pragma unroll
    for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) fold[wave - 1][(cbk * NCB + cb) * 4 + e][lane] = acc[cbk][cb][e];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int cbk = 0; cbk < CBW; ++cbk)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[cbk][cb][e] += fold[w][(cbk * NCB + cb) * 4 + e][lane];
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
// Parameter gradients from the row-split partials of Q [splits][C_p + 1][JP].  One workgroup per kTailPc channel rows: it sums
// its rows' partials in split order (the fixed-order reduction, no separate pass), stages them (transposed) next to Bm and its
// rows of W_p in LDS, then two small register-tiled products:
//   dW_p[c][n]        = sum_{j <= F} Q[c][j] Bm[j][n]                 thread tile 4 rows x 4 columns (16-byte LDS reads of both)
//   partB[blk][(n,f)] = sum_{c in blk} Q[c][1+f] W_p[c][n]            thread tile 4 columns n x 4 fields (-> dwsum_L, reduced by
//                                                                     cin_tail_fill_kernel)
//   workgroup 0: dbias_p[n] = sum_j Q[C_p][j] Bm[j][n], and the reduced ones row Q[C_p][.] -> qones (cin_tail_fill_kernel)
constexpr int kTailPc = 32;
static __global__ __launch_bounds__(256) void cin_tail_params_kernel(const float* __restrict__ Qpart, int splits, const float* __restrict__ Wp,
                                                              const float* __restrict__ bmT, float* __restrict__ dWp,
                                                              float* __restrict__ dbiasp, float* __restrict__ partB,
                                                              float* __restrict__ qones, int Cp, int F, int Hq, int JP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // Bm [F+1][ldb] | Q^T [JP][kTailPc + 4] | W_p rows [kTailPc][ldb]
  const int J1 = F + 1;
  const int ldb = (Hq + 3) & ~3;            // 16-byte aligned rows
  constexpr int ldq = kTailPc + 4;
  float* bm = smem;
  float* qt = smem + (size_t)J1 * ldb;      // qt[j][cl]: the workgroup's rows of Q, transposed; column kTailPc: the ones row
  float* wps = qt + (size_t)JP * ldq;
  // (two workgroups per block of rows: the even one computes dW_p / dbias_p, the odd one the dwsum_L partials -- twice the
  // workgroups in flight for a kernel whose ~150 workgroups would otherwise leave 100 CUs idle)
  const int phase = blockIdx.x & 1, blk = blockIdx.x >> 1;
  const int c0 = blk * kTailPc;
  const int nc = min(kTailPc, Cp - c0);
  const long qstride = (long)(Cp + 1) * JP;
  for (int idx = threadIdx.x; idx < (kTailPc + 1) * JP; idx += 256) {
    const int cl = idx / JP, j = idx - cl * JP;
    const int c = cl < kTailPc ? c0 + cl : Cp;   // last staged row: the ones channel
    float t0 = 0.f, t1 = 0.f;
    if (cl == kTailPc || cl < nc) {
      const float* src = Qpart + (long)c * JP + j;
      int sp = 0;
      for (; sp + 1 < splits; sp += 2) {
        t0 += src[(long)sp * qstride];
        t1 += src[(long)(sp + 1) * qstride];
      }
      if (sp < splits) t0 += src[(long)sp * qstride];
    }
    qt[j * ldq + cl] = t0 + t1;
  }
  if (phase == 0) {
    for (int idx = threadIdx.x; idx < J1 * ldb; idx += 256) {
      const int j = idx / ldb, n = idx - j * ldb;
      bm[idx] = n < Hq ? (j == 0 ? 1.f : bmT[(j - 1) * Hq + n]) : 0.f;
    }
  } else {
    for (int idx = threadIdx.x; idx < kTailPc * ldb; idx += 256) {
      const int cl = idx / ldb, n = idx - cl * ldb;
      wps[idx] = (cl < nc && n < Hq) ? Wp[(long)(c0 + cl) * Hq + n] : 0.f;
    }
  }
  __syncthreads();
  // dW_p: tiles of 4 rows x 4 columns
  const int nq = ldb >> 2;                  // column quads
  if (phase == 0)
  for (int tile = threadIdx.x; tile < (kTailPc / 4) * nq; tile += 256) {
    const int cq = tile / nq, n4 = tile - cq * nq;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int j = 0; j < J1; ++j) {
      const float4 qv = *reinterpret_cast<const float4*>(qt + j * ldq + 4 * cq);
      const float4 bv = *reinterpret_cast<const float4*>(bm + j * ldb + 4 * n4);
      const float qa[4] = {qv.x, qv.y, qv.z, qv.w}, ba[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(qa[a], ba[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cl = 4 * cq + a;
      if (cl < nc) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (4 * n4 + b < Hq) dWp[(long)(c0 + cl) * Hq + 4 * n4 + b] = acc[a][b];
      }
    }
  }
  if (blockIdx.x == 0) {
    for (int n = threadIdx.x; n < Hq; n += 256) {
      float t = 0.f;
      for (int j = 0; j < J1; ++j) t = fmaf(qt[j * ldq + kTailPc], bm[j * ldb + n], t);
      dbiasp[n] = t;
    }
    for (int j = threadIdx.x; j < JP; j += 256) qones[j] = qt[j * ldq + kTailPc];
  }
  if (phase == 0) return;
  // partB: tiles of 4 columns n x 4 fields, both operands from LDS
  constexpr int kTailFt = 4;
  const int nfg = (F + kTailFt - 1) / kTailFt;
  float* pb = partB + (long)blk * Hq * F;
  for (int tile = threadIdx.x; tile < nq * nfg; tile += 256) {
    const int fg = tile / nq, n4 = tile - fg * nq;
    float acc[kTailFt][4];
#pragma unroll
    for (int a = 0; a < kTailFt; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const float* qrow[kTailFt];
#pragma unroll
    for (int a = 0; a < kTailFt; ++a) qrow[a] = qt + min(1 + kTailFt * fg + a, F) * ldq;
#pragma unroll 4
    for (int cl = 0; cl < kTailPc; ++cl) {   // (rows past nc are zero in both images)
      const float4 w4 = *reinterpret_cast<const float4*>(wps + cl * ldb + 4 * n4);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int a = 0; a < kTailFt; ++a) {
        const float qv = qrow[a][cl];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(qv, wv[b], acc[a][b]);
      }
    }
#pragma unroll
    for (int a = 0; a < kTailFt; ++a) {
      const int f = kTailFt * fg + a;
      if (f < F) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (4 * n4 + b < Hq) pb[(4 * n4 + b) * F + f] = acc[a][b];
      }
    }
  }
}

// dwsum_L[(n,f)] = sum_blk partB[blk][(n,f)] + Q[C_p][1+f] bias_p[n];  dW_L[(n,f)][n'] = dwsum_L[(n,f)] for every n';
// workgroup 0: dbias_L[n'] = Q[C_p][F+1] (= sum_m dP_L[m]).  qones = the reduced ones row Q[C_p][.] (cin_tail_params_kernel).  64 rows per workgroup, partial sums folded through LDS in fixed order.
static __global__ __launch_bounds__(256) void cin_tail_fill_kernel(const float* __restrict__ partB, int parts, const float* __restrict__ qones,
                                                            const float* __restrict__ biasp, float* __restrict__ dWL,
                                                            float* __restrict__ dbiasL, int F, int Hq, int HL) {
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
    val[rl] = ((red[0][rl] + red[1][rl]) + (red[2][rl] + red[3][rl])) + qones[1 + f] * biasp[n];
  }
  __syncthreads();
  const int nrow = min(64, rows - r0);
  for (int idx = threadIdx.x; idx < nrow * HL; idx += 256) {
    const int r = idx / HL;
    dWL[(long)r0 * HL + idx] = val[r];
  }
  if (blockIdx.x == 0) {
    const float s = qones[F + 1];
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
// KS > 1 (small M): KS waves of the workgroup share one block of 32 rows and split the PERIODS (values of h) between them; each
// writes the G^{p-1} columns of its own h range, the dX partial sums meet in the LDS staging area.
template <int JT, int NQ, int SMODE = 0, int KS = 1>
__global__ __launch_bounds__(256, 2) void cin_tail_dz_kernel(const float* __restrict__ Uz, const float* __restrict__ xT,
                                                              const float* __restrict__ xpT, int xps, const float* __restrict__ Y, int JP,
                                                              const float* __restrict__ dP, int ldp, int K, int lp, int lL,
                                                              float* __restrict__ GprevT, int HSp, float* __restrict__ dxT, int M, int F,
                                                              int Hp, int periods) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // dX staging [JT][256] | per-wave line buffers [32][kGlStride]
  constexpr int JHp = 4 * NQ;
  constexpr int P = JT / gcd_c(16, JT);
  constexpr int HPP = 16 * P / JT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int kpart = wave % KS;
  const int wrow0 = ((blockIdx.x * 4 + wave) / KS) * 32;
  if (KS == 1 && wrow0 >= M) return;
  // this wave's periods: [per_lo, per_hi); whole 32-column lines of G^{p-1} per wave (the line buffer is flushed line by line)
  constexpr int kLinePer = HPP >= 32 ? 1 : 32 / HPP;
  const int pq = ((periods + KS - 1) / KS + kLinePer - 1) / kLinePer * kLinePer;
  const int per_lo = KS == 1 ? 0 : min(periods, kpart * pq);
  const int per_hi = KS == 1 ? periods : (wrow0 < M ? min(periods, per_lo + pq) : per_lo);
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
      dxa[j] = (KS == 1 || kpart == 0) ? dpl * __builtin_bit_cast(float, __builtin_bit_cast(int, yt[j]) & keep) : 0.f;   // (the direct term: once)
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
  for (int s4 = 0; s4 < NQ; ++s4) q[s4] = wz[(long)per_lo * P * kTileStride + s4];
  float gx = 0.f;
  f32x16 dprev;
#pragma unroll
  for (int e = 0; e < 16; ++e) dprev[e] = 0.f;
  float xprev[HPP], xcur[HPP], gout[HPP];
#pragma unroll
  for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl] = gout[hl] = 0.f;
  int hprev = -HPP;
  // (the prologue's loads are waited for once, here: merged with them the wait count at the top of the period loop collapses
  // to vmcnt(0), i.e. the A-operand prefetch of the next tile is drained at every period -- see cin_fwd3_kernel)
#pragma unroll
  for (int s4 = 0; s4 < NQ; ++s4) settle(q[s4].x);
#pragma unroll
  for (int sx = 0; sx < JHp; ++sx) settle(areg[sx]);

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
  for (int per = per_lo; per < per_hi; ++per) {
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
        auto slots = [&](int lo, int hi) {
#pragma unroll
          for (int sl = lo; sl < hi; ++sl) {
            if (tp == 0) slot_apply(dprev, xprev, hprev, P - 1, sl);
            else slot_apply(dprev, xcur, hbase, tp - 1, sl);
          }
        };
        // the previous tile's 16 slots: SMODE 0 spread over this tile's NQ step groups (behind each group's 4 MFMAs);
        // 1 all behind the first group; 2 all in front of the tile's first MFMA; 3 one slot behind each of the first 16 MFMAs
        if (SMODE == 2 && s4 == 0) {
          slots(0, 16);
          __builtin_amdgcn_sched_barrier(0);
        }
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d = mfma32(wv[e], areg[4 * s4 + e], d);
          if (SMODE == 3 && 4 * s4 + e < 16) {
            __builtin_amdgcn_sched_barrier(0);
            slots(4 * s4 + e, 4 * s4 + e + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (SMODE == 0) slots((16 * s4) / NQ, (16 * (s4 + 1)) / NQ);
        if (SMODE == 1 && s4 == 0) slots(0, 16);
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
  if constexpr (KS > 1) {
    __syncthreads();
    if (kpart > 0 || wrow0 >= M) return;
  } else {
    __builtin_amdgcn_wave_barrier();
  }
  const float* dsc = smem + wave * 64;
  const int nrow = min(32, M - wrow0);
  float* dst = dxT + (long)wrow0 * F;
  for (int idx = lane; idx < nrow * F; idx += 64) {
    const int rr = idx / F, f = idx - rr * F;
    float v = dsc[(f >> 1) * 256 + (f & 1) * 32 + rr];
#pragma unroll
    for (int k = 1; k < KS; ++k) v += dsc[(f >> 1) * 256 + (f & 1) * 32 + rr + 64 * k];   // the group's partial sums, in wave order
    dst[idx] = v;
  }
}

}  // namespace fil
