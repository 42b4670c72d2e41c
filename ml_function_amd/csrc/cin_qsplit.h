// Merged quadratic tail on SPLIT-bf16 operands ("bf16x3", mode bit FIL_CIN_BF16X3): the labelled reduced-operand mode of SURVEY 8 A3
// (reference interactive_layer.py:310-327) for the three GEMM launches of cin_qmerge.h.  The small kernels stay exact fp32.
//
// Every fp32 operand v is cut into three bf16 pieces v = v1 + v2 + v3 -- v1 = the top 8 significand bits of v (truncated), v2 = the top
// 8 bits of v - v1, v3 = v - v1 - v2: the cut is EXACT, a 24-bit significand is three 8-bit ones -- and a product a b is accumulated as
//   a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1      (small terms first; the three dropped terms are below 2^-24 |a b|)
// by six v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the bf16 matrix pipe runs at 16x the rate of v_mfma_f32_32x32x2_f32, so the
// same reduction costs 6 x 32 cycles per 16 reduction indices instead of 8 x 64, and -- unlike the f32-input MFMA, which occupies the
// vector ALU -- vector instructions issue in its shadow (MI355X_MICROARCH.md: ~5 per MFMA), which is where the cuts are made.
// A bf16 piece keeps fp32's exponent range, so there is no scaling; what differs from the exact kernels: the accumulation order
// (six partial products per term), and an infinite operand gives NaN (inf - inf in its second piece) where the exact chain gives inf.
//
// MFMA 32x32x16 operand maps: lane (r = lane & 31, half = lane >> 5) supplies A[i = r][k = 8 half .. 8 half + 7] and
// B[k = 8 half .. 8 half + 7][j = r], eight bf16 in four dwords; the accumulator layout is the 32x32x2 one (mfma32_row).
#pragma once
#include "cin_qmerge.h"

namespace fil {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32b(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A B to fp32 accuracy from the pieces, small terms first
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  c = mfma32b(a[2], b[0], c);
  c = mfma32b(a[0], b[2], c);
  c = mfma32b(a[1], b[1], c);
  c = mfma32b(a[1], b[0], c);
  c = mfma32b(a[0], b[1], c);
  c = mfma32b(a[0], b[0], c);
  return c;
}

// two fp32 values -> their top halves in one dword (element 0 in the low half): a bf16 pair by truncation
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float top8(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
// eight fp32 values -> the three bf16 planes of one MFMA operand (6.5 vector instructions per value: and, sub, and, sub + 1.5 perm)
__device__ __forceinline__ void split3(const float (&p)[8], u32x4 (&a)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float p0 = p[2 * q], p1 = p[2 * q + 1];
    const float r0 = p0 - top8(p0), r1 = p1 - top8(p1);
    const float s0 = r0 - top8(r0), s1 = r1 - top8(r1);
    a[0][q] = pack_hi(p0, p1);
    a[1][q] = pack_hi(r0, r1);
    a[2][q] = pack_hi(s0, s1);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Geometry of the pair slots.  A wave half walks the slots s = h JT + j (the exact kernels' steps: pair (h, (h + 2j + half) mod F)) eight
// per MFMA; a PERIOD is HPS values of h = KP whole MFMA steps (HPS JT = 8 KP, KP >= 2), inside which slot -> (h - h0, j) is compile-time.
template <int JT>
struct QsGeo {
  static constexpr int HPS0 = 8 / gcd_c(8, JT);
  static constexpr int HPS = HPS0 * JT / 8 >= 2 ? HPS0 : 2 * HPS0;
  static constexpr int KP = HPS * JT / 8;
  static constexpr int WS = HPS + 2 * JT - 2;   // window of wrapped positions a period touches: t = (h - h0) + 2j
  static_assert(HPS * JT == 8 * KP && KP >= 2, "a period is a whole number (>= 2) of 8-slot steps");
};
inline int cin_qs_hps(int JT) {
  const int h0 = 8 / cin_gcd(8, JT);
  return h0 * JT / 8 >= 2 ? h0 : 2 * h0;
}
inline int cin_qs_steps(int F, int JT) {   // MFMA steps of the forward's reduction (slots past F JT carry zero weights)
  const int hps = cin_qs_hps(JT);
  return (F + hps - 1) / hps * (hps * JT / 8);
}
constexpr int kQsStageBytes = 24 * 1024;   // one forward step of B planes: [plane 3][column block 8][lane 64][8 bf16]
constexpr int kQsStages = 4;

// [W1s | Ts] in the forward operand layout of the exact kernel ([slot s][half][r][4]: cin_pack_wf_sym_body) -> the split planes
// Wb [step t][plane][nb 0..7][lane][8 bf16]: element e of lane (r, half) = the weight of slot 8 t + e, column 4 r + (nb & 3) of
// W1s (nb < 4) or Ts (nb >= 4).  One thread per (t, nb, lane).
static __global__ __launch_bounds__(256) void cin_qs_pack_wb_kernel(const float* __restrict__ W1f, const float* __restrict__ WTf, u32x4* __restrict__ Wb,
                                                                    int NT, int nslots) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= NT * 512) return;
  const int lane = idx & 63, nb = (idx >> 6) & 7, t = idx >> 9;
  const int r = lane & 31, half = lane >> 5;
  const float* src = (nb < 4 ? W1f : WTf) + half * 128 + 4 * r + (nb & 3);
  float p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int s = 8 * t + e;
    p[e] = s < nslots ? src[(long)s * 256] : 0.f;
  }
  u32x4 a[3];
  split3(p, a);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) Wb[((long)(t * 3 + pl) * 8 + nb) * 64 + lane] = a[pl];
}

// ------------------------------------------------------------------------------------------------------------------------------
// Forward: [x1 | R] = pairs(x) [W1s | Ts] on split operands.  Workgroup = 8 waves = 256 rows (two waves per SIMD), wave = 32 rows x 256
// columns as in cin_fwdq_kernel; everything behind the main loop IS that kernel's epilogue (cin_fwdq_epilogue).
//   B (weights): the 24 KB of planes of a step are the same for every wave -- streamed ONCE per workgroup into a ring of kQsStages
//     LDS slots by LDS-DMA (buffer_load ... lds: no registers; each wave issues three 1-KB pieces of a step, three steps ahead), one
//     workgroup barrier per step; a wave reads its operands with ds_read_b128 (conflict-free: [lane][16 B]).  Streaming them through
//     registers instead is 24 KB per wave and step: 64 B/clk/CU at two waves per SIMD, the whole L1 bandwidth.
//   A (generated): the lane keeps a sliding window of its row's wrapped positions (x2T, cin_transpose_in_body) over the period's HPS
//     values of h -- product, cut, pack: ~60 vector instructions per step, issued in the shadow of the step's 48 MFMAs, one step ahead.
template <int JT>
__global__ __launch_bounds__(512, 2) void cin_fwdq_b_kernel(const float* __restrict__ x2T, int XL, const u32x4* __restrict__ Wb, int NT,
                                                            const float* __restrict__ bias1, const float* __restrict__ wsn, int JTG,
                                                            const float* __restrict__ cvec, float* __restrict__ x1T, float* __restrict__ RT, int HS,
                                                            float* __restrict__ pool1, float* __restrict__ pool_p, float* __restrict__ pool_L, int M, int F,
                                                            int H, CinHeadFold hf) {
  using G = QsGeo<JT>;
  constexpr int HPS = G::HPS, KP = G::KP, WS = G::WS, NS = kQsStages, SB = kQsStageBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];   // [NS][SB]
  __shared__ float lin_s[8][32];
  __shared__ float pv_s[8][3][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 8 + wave) * 32;
  const bool active = wrow0 < M;   // (a wave past the end still takes part in the ring: DMA pieces, barriers)
  const int wrow_u = __builtin_amdgcn_readfirstlane(wrow0);
  // this wave's half of its 64-row block of the wrapped rows ([p][64 rows]); an inactive wave reads zeros
  const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(x2T + (active ? (long)(wrow_u >> 6) * XL * 64 : 0), active ? (long)XL * 256 : 0);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(reinterpret_cast<const float*>(Wb), (long)NT * SB);   // steps past the end read zeros
  const int wo = (half * 32 + r) * 16;
  const int vrow = ((wrow_u & 63) + r) * 4, vhalf = vrow + half * 256;
  auto ldx = [&](int voff, int p) {   // wrapped position p (uniform) of the lane's row
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, __builtin_amdgcn_readfirstlane(p) * 256, 0));
  };
  auto dma = [&](int t) {   // this wave's three pieces of step t -> ring slot t % NS
    unsigned char* dst = ring + (t % NS) * SB + wave * 3072;
    const int so = __builtin_amdgcn_readfirstlane(t * SB + wave * 3072);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, lane * 16, so + q * 1024, 0, 0);
  };

  f32x16 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  // window of period 0: wl[t] = x2[half + t], xp[u] = x[m, u]
  float wl[WS], xp[HPS], wn[HPS], xn[HPS];
#pragma unroll
  for (int t = 0; t < WS; ++t) wl[t] = ldx(vhalf, t);
#pragma unroll
  for (int u = 0; u < HPS; ++u) {
    const float v = ldx(vrow, u);
    xp[u] = u < F ? v : 0.f;
  }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) dma(s);
  // A operand of a step of the current period / of the first step of the next one
  auto make_a = [&](int kk, u32x4 (&a)[3]) {
    float p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int st = 8 * kk + e;   // compile-time
      p[e] = xp[st / JT] * wl[st / JT + 2 * (st % JT)];
    }
    split3(p, a);
  };
  auto make_a_next = [&](const float (&xnm)[HPS], u32x4 (&a)[3]) {
    float p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = HPS + e / JT + 2 * (e % JT);   // the next period's window is this one shifted by HPS, its tail = wn
      p[e] = xnm[e / JT] * (t < WS ? wl[t < WS ? t : 0] : wn[t >= WS ? t - WS : 0]);
    }
    split3(p, a);
  };
  u32x4 acur[3], anext[3];
  make_a(0, acur);
  const int nper = NT / KP;
  const int ldsb = lane * 16;
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // step 0's pieces (the in-loop wait of a period's first step allows for window loads
                                                     // that the very first period has not issued)
#pragma unroll 1
  for (int per = 0; per < nper; ++per) {
    const int h0 = per * HPS;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int t = per * KP + kk;
      // Step t's pieces were issued three steps ago; younger vector-memory operations may stay in flight: the DMA pieces of steps
      // t+1, t+2 (6) and, for the two steps after a period's first one, the window loads issued there in front of its DMA (2 HPS).
      if (kk == 1 || kk == 2 || (kk == 0 && KP == 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + 2 * HPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave's pieces of step t have landed, and every wave is done reading slot (t - 1) % NS
      if (kk == 0) {
        // the next period's x[m, h] values and the HPS window entries that slide in (positions past XL / rows of an inactive wave: zeros)
#pragma unroll
        for (int u = 0; u < HPS; ++u) {
          xn[u] = ldx(vrow, h0 + HPS + u);     // (h >= F is masked where the value is USED: a select here would wait for the load)
          wn[u] = ldx(vhalf, h0 + WS + u);
        }
      }
      dma(t + NS - 1);   // (past NT: zeros into a free slot -- keeps the count of operations in flight the same in every step)
      if (kk + 1 < KP) make_a(kk + 1, anext);
      else {
        // x[m, h] of the padded h >= F: zero (their weights are zero, but the wrapped row holds x[m, h - F] there: NaN x 0)
#pragma unroll
        for (int u = 0; u < HPS; ++u) xn[u] = h0 + HPS + u < F ? xn[u] : 0.f;
        make_a_next(xn, anext);
      }
      const unsigned char* sb = ring + (t % NS) * SB + ldsb;
      u32x4 b[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[0][pl] = *reinterpret_cast<const u32x4*>(sb + (pl * 8) * 1024);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        if (nb + 1 < 8) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[(nb + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(sb + (pl * 8 + nb + 1) * 1024);
        }
        acc[nb] = mfma_split(acur, b[nb & 1], acc[nb]);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) acur[pl] = anext[pl];
    }
    // slide the window
#pragma unroll
    for (int t = 0; t + HPS < WS; ++t) wl[t] = wl[t + HPS];
#pragma unroll
    for (int u = 0; u < HPS; ++u) {
      if (WS - HPS + u >= 0) wl[WS - HPS + u] = wn[u];
      xp[u] = xn[u];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the trailing DMA pieces: nothing may land in LDS after the workgroup is gone)
  if (!active) return;   // (no workgroup barriers below)
  cin_fwdq_epilogue<8>(acc, rx, vhalf, wo, r, half, wave, wrow0, bias1, wsn, JTG, cvec, x1T, RT, HS, pool1, pool_p, pool_L, M, F, H, hf, lin_s, pv_s);
}

bool cin_launch_fwdq_b(hipStream_t st, int JT, const float* x2T, int XL, const u32x4* Wb, int NT, const float* bias1, const float* wsn, int JTG,
                       const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf);

}  // namespace fil
