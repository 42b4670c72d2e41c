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
#include "cin_qmerge.h"   // (-> cin_qtail.h -> cin_split.h: the cut, the MFMA helpers, the plane layouts and their pack bodies)

namespace fil {

// ------------------------------------------------------------------------------------------------------------------------------
// Forward: [x1 | R] = pairs(x) [W1s | Ts] on split operands.  Workgroup = 8 waves = 256 rows (two waves per SIMD), wave = 32 rows x 256
// columns as in cin_fwdq_kernel; everything behind the main loop IS that kernel's epilogue (cin_fwdq_epilogue).
//   B (weights): the 24 KB of planes of a step are the same for every wave -- streamed ONCE per workgroup into a ring of kQsStages
//     LDS slots by LDS-DMA (buffer_load ... lds: no registers; each wave issues three 1-KB pieces of a step, three steps ahead), one
//     workgroup barrier per step; a wave reads its operands with ds_read_b128 (conflict-free: [lane][16 B]).  Streaming them through
//     registers instead is 24 KB per wave and step: 64 B/clk/CU at two waves per SIMD, the whole L1 bandwidth.
//   A (generated): the lane keeps a sliding window of its row's wrapped positions (x2T, cin_transpose_in_body) over the period's HPS
//     values of h -- product, cut, pack: ~60 vector instructions per step, issued in the shadow of the step's 48 MFMAs, one step ahead.
template <int JT, int NW>   // NW waves per workgroup: 8 (one workgroup per CU, ring of 4) or 4 (two independent ones per CU, rings of 3)
__global__ __launch_bounds__(64 * NW, 2) void cin_fwdq_b_kernel(const float* __restrict__ x2T, int XL, const u32x4* __restrict__ Wb, int NT,
                                                            const float* __restrict__ bias1, const float* __restrict__ wsn, int JTG,
                                                            const float* __restrict__ cvec, float* __restrict__ x1T, float* __restrict__ RT, int HS,
                                                            float* __restrict__ pool1, float* __restrict__ pool_p, float* __restrict__ pool_L, int M, int F,
                                                            int H, CinHeadFold hf) {
  using G = QsGeo<JT>;
  constexpr int HPS = G::HPS, KP = G::KP, WS = G::WS, NS = NW == 8 ? kQsStages : 3, SB = kQsStageBytes;
  constexpr int PW = 24 / NW;                                            // 1-KB DMA pieces of a step per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];   // [NS][SB]
  __shared__ float lin_s[NW][32];
  __shared__ float pv_s[NW][3][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * NW + wave) * 32;
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
  auto dma = [&](int t) {   // this wave's PW pieces of step t -> ring slot t % NS
    unsigned char* dst = ring + (t % NS) * SB + wave * (PW * 1024);
    const int so = __builtin_amdgcn_readfirstlane(t * SB + wave * (PW * 1024));
#pragma unroll
    for (int q = 0; q < PW; ++q)
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
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PW) : "memory");   // step 0's pieces (the in-loop wait of a period's first step allows
                                                                         // for window loads that the very first period has not issued)
#pragma unroll 1
  for (int per = 0; per < nper; ++per) {
    const int h0 = per * HPS;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int t = per * KP + kk;
      // Step t's pieces were issued NS-1 steps ago; younger vector-memory operations may stay in flight: the DMA pieces of the NS-2
      // steps behind it and, for the NS-2 steps after a period's first one, the window loads issued there in front of its DMA (2 HPS).
      const int since = kk == 0 ? KP : kk;   // steps since the last window loads (compile-time once the step loop is unrolled)
      if (since <= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PW + 2 * HPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PW) : "memory");
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
  cin_fwdq_epilogue<NW>(acc, rx, vhalf, wo, r, half, wave, wrow0, bias1, wsn, JTG, cvec, x1T, RT, HS, pool1, pool_p, pool_L, M, F, H, hf, lin_s, pv_s);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Data gradients (cin_dz2_kernel on split operands): dZ^T tile = 32 slot rows (A: the slot-ordered pair weights, three planes) x 32 rows m
// (B: the lane's half row of G1 / dP_L x1, cut ONCE per pass into 8 steps x 3 planes = 96 registers), 8 MFMA steps of 16 columns =
// 48 bf16 MFMAs per tile instead of 64 f32 ones.  The accumulator layout, the slot order and the whole register / LDS contraction of
// the tile into the dX image are cin_dz2_kernel's, unchanged: per tile the same ~140 vector / LDS instructions now stand beside
// 1536 cycles of matrix time instead of 4096 -- they issue in the MFMAs' shadow.  The A planes stream through registers (a queue QD
// steps deep, 16-byte scalar-offset buffer loads): 24 KB per tile and wave -- the LDS holds the two workgroups' dX images, there is no
// room for a shared ring, so this kernel is bound by the L1 (64 B/clk/CU = its MFMA time at two waves per SIMD).
//
// Wzb [tile][step t][plane][lane][8 bf16]: cin_qs_pack_wz_body (cin_split.h).
template <int JT, int G>
__global__ __launch_bounds__(256, 2) void cin_dz2_b_kernel(const float* __restrict__ g1T, const float* __restrict__ g2T, int HS,
                                                           const float* __restrict__ dsc, int ldp, int K, const u32x4* __restrict__ Wzb1,
                                                           const u32x4* __restrict__ Wzb2, const float* __restrict__ xT, float* __restrict__ dxT,
                                                           int accumulate, int M, int F, int H1, int H2, int periods, int FR,
                                                           float* __restrict__ dx, const float* __restrict__ cvec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [FR][128 rows][2]
  constexpr int P = JT / gcd_c(16, JT);
  constexpr int HPP = 16 * P / JT;
  constexpr int NT = 8;                                // MFMA steps per tile (16 columns each: 8 per wave half)
  constexpr int FS = kDz2FieldStride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M) return;   // (no workgroup barriers in this kernel)
  float* lrow = smem + (wave * 32 + r) * 2;            // this lane's row: field f at lrow[f*FS + {0: x, 1: dX}]
  const int m = wrow0 + r;
  const bool vq = m < M;
  const long mq = vq ? m : M - 1;
  f32x4s gq[16];
  {
    const f32x4s* grow4 = reinterpret_cast<const f32x4s*>(g1T + mq * HS + half * 64);
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) gq[s4] = grow4[s4];
  }
  for (int f0 = half; f0 < FR; f0 += 16) {
    float xt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xt[u] = xT[mq * F + min(f0 + 2 * u, F - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + 2 * u;
      if (f < FR) {
        const int keep = (vq && f < F) ? -1 : 0;
        *reinterpret_cast<float2*>(lrow + f * FS) = make_float2(__builtin_bit_cast(float, __builtin_bit_cast(int, xt[u]) & keep), 0.f);
      }
    }
  }
  float dpl;
  {
    const long bb = mq / K;
    dpl = dsc[bb * ldp + (mq - bb * K)];
  }
  __builtin_amdgcn_wave_barrier();
  const long wbytes = ((long)periods * P + 1) * NT * 3 * 1024;
  const int wo = lane * 16;
  // B operand of the current pass: the lane's half row (columns past the layer's width and rows past M zeroed; pass 1: scaled by dP_L),
  // cut into the planes of the tile's eight steps
  u32x4 gpl[NT][3];
  auto to_gpl = [&](const f32x4s (&gv4)[16], int Hk, float sc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float p[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int keep = (vq && half * 64 + 8 * t + e < Hk) ? -1 : 0;
        const float gv = gv4[2 * t + (e >> 2)][e & 3];   // (a copy: __builtin_bit_cast on the vector ELEMENT expression reads element 0 for every e)
        p[e] = __builtin_bit_cast(float, __builtin_bit_cast(int, gv) & keep) * sc;
      }
      split3(p, gpl[t]);
    }
  };
  to_gpl(gq, H1, 1.f);

#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const float* gT = pass == 0 ? g1T : g2T;
    const int Hk = pass == 0 ? H1 : H2;
    const float sc = pass == 0 ? 1.f : dpl;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const float*>(pass == 0 ? Wzb1 : Wzb2), wbytes);
    auto ldw = [&](int t, int st, int pl) {   // step st of tile t
      return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wo + pl * 1024, (t * NT + st) * 3072, 0));
    };
    constexpr int QD = 4;   // queue depth in steps (half a tile)
    u32x4 q[QD][3];
#pragma unroll
    for (int st = 0; st < QD; ++st)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) q[st][pl] = ldw(0, st, pl);
    if (pass == 1) {
      const f32x4s* grow4 = reinterpret_cast<const f32x4s*>(gT + mq * HS + half * 64);
      f32x4s g2[16];
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) g2[s4] = grow4[s4];
      to_gpl(g2, Hk, sc);
    }
    float gx = 0.f;
    f32x16 dprev;
#pragma unroll
    for (int i = 0; i < 16; ++i) dprev[i] = 0.f;
    float xprev[HPP], xcur[HPP];
#pragma unroll
    for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl] = 0.f;
    int hprev = 0;
    float2 lv[G];
    float* la[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      lv[k] = make_float2(0.f, 0.f);
      la[k] = lrow;
    }
    float *abase = lrow, *awrap = lrow;
    int symh = 0;
    auto sym_period = [&](int hb) {
      symh = hb + half;
      abase = lrow + symh * FS;
      awrap = abase - F * FS;
    };
    auto slot_fetch = [&](int tp, int rr, int k) {
      const int sp = 16 * tp + rr;
      const int off = sp / JT + 2 * (sp % JT);
      la[k] = (symh >= F - off ? awrap : abase) + off * FS;
      lv[k] = *reinterpret_cast<const float2*>(la[k]);
    };
    auto slot_apply = [&](const f32x16& d, const float (&xpv)[HPP], int hb, int tp, int rr, int k) {
      const int sp = 16 * tp + rr;
      const int hl = sp / JT, j = sp % JT;
      const float dz = d[rr];
      gx = fmaf(dz, lv[k].x, gx);
      la[k][1] = fmaf(dz, xpv[hl], lv[k].y);
      if (j == JT - 1) {
        const float t = lane_halves_sum(gx);
        gx = 0.f;
        if (half == 0) __hip_atomic_fetch_add(lrow + (hb + hl) * FS + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    };
    sym_period(hprev);
#pragma unroll
    for (int k = 0; k < G; ++k) slot_fetch(P - 1, k, k);
    // slots per step: the previous tile's 16 slots are contracted over this tile's 8 steps, in blocks of G behind every G/2 steps
    constexpr int SPB = G >= 2 ? G / 2 : 1;      // steps per block of G slots (G = 1: one slot per half step is not expressible: two per step)
    static_assert(G == 4 || G == 2, "blocks of two or four slots");
#pragma unroll 1
    for (int per = 0; per < periods; ++per) {
      const int hbase = per * HPP;
#pragma unroll
      for (int hl = 0; hl < HPP; ++hl) xcur[hl] = lrow[(hbase + hl) * FS];
#pragma unroll
      for (int tp = 0; tp < P; ++tp) {
        const int t = per * P + tp;
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int st = 0; st < NT; ++st) {
          d = mfma_split(q[st % QD], gpl[st], d);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) q[st % QD][pl] = st + QD < NT ? ldw(t, st + QD, pl) : ldw(t + 1, st + QD - NT, pl);
          if (st % SPB == SPB - 1) {
            const int blk = st / SPB;   // block of G slots: blk*G .. blk*G + G-1
#pragma unroll
            for (int k = 0; k < G; ++k) {
              const int rr = blk * G + k;
              if (tp == 0) slot_apply(dprev, xprev, hprev, P - 1, rr, k);
              else slot_apply(dprev, xcur, hbase, tp - 1, rr, k);
            }
            if (blk * G + G < 16) {
#pragma unroll
              for (int k = 0; k < G; ++k) slot_fetch(tp == 0 ? P - 1 : tp - 1, blk * G + G + k, k);
            } else {
              if (tp == 0) sym_period(hbase);
#pragma unroll
              for (int k = 0; k < G; ++k) slot_fetch(tp, k, k);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        dprev = d;
      }
#pragma unroll
      for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl];
      hprev = hbase;
    }
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += G) {
#pragma unroll
      for (int k = 0; k < G; ++k) slot_apply(dprev, xprev, hprev, P - 1, r0 + k, k);
      if (r0 + G < 16) {
#pragma unroll
        for (int k = 0; k < G; ++k) slot_fetch(P - 1, r0 + G + k, k);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  cin_dz2_finish(smem, wave, lane, wrow0, M, F, K, dsc, ldp, dxT, accumulate, dx, cvec);
}

bool cin_launch_dz2_b(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const u32x4* Wzb1,
                      const u32x4* Wzb2, const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods, float* dx,
                      const float* cvec);

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradients (cin_dwq_kernel's GEMM on split operands): [dW1s | dTs | v^T] = P^T [G1 | x1], reduction over the rows m, 16 per MFMA
// step.  Wave = 32 channel rows x 256 columns as before.  Operands:
//   A (generated, channel rows on the lanes): A[c][m] = xe[m,h_c] xe[m,f_c] (columns 0..127), A' = A * xe[m, F+1+sel_c] (128..255) for the
//     step's 16 rows -- the rows' xe entries come into LDS by LDS-DMA (2.7 KB per step), the lane gathers its three columns there;
//   B ([G1 | x1], fp32 row-major in memory): a lane needs EIGHT ROWS of one column per plane vector, and cutting a 16 x 256 tile is
//     ~350 vector instructions -- too many for one wave per step, so the four waves of a half workgroup, which take four channel tiles
//     of ONE row split, cut each step's tile together (each thread: 2 x 8 values, a column of G1 and one of x1) one step ahead and
//     hand the planes round through LDS (ring of two 24-KB slots; operands read by ds_read_b128 as in the forward).
// Step index k = 8 half + e of the MFMA stands for row 16 t + rho(half, e), rho = (e & 3) + 8 (e >> 2) + 4 half (any bijection works
// as long as A and B agree; this one is the accumulator row order, mfma32_row, of the kernels that produce G1 and x1).
// Work: item = (row split, group of 4 channel tiles) = one workgroup of 4 waves, TWO workgroups per CU (60 KB of LDS each).  They
// are independent (a barrier per step inside each), so their steps drift apart and one's vector phase -- the cut of the next B tile,
// the A planes: ~250 instructions between the barrier and the first MFMA -- runs under the other's MFMAs on the same SIMDs (as ONE
// 8-wave workgroup sharing the barrier the two waves of a SIMD were in step and the matrix pipe idled through that phase: 178 us).
// 26 tiles at F = 39 are 6 groups of four and one of two (2 of 28 wave slots idle).  No fold of split pairs (the LDS is the rings'):
// one partial [C][256] per row split, summed by cin_reduce_expand_q_kernel as before.
constexpr int kDwqbPlaneSlot = 24 * 1024, kDwqbXeSlot = 4096;
constexpr int kDwqbHalfBytes = 2 * kDwqbPlaneSlot + 3 * kDwqbXeSlot;   // planes ring (2) + xe ring (3) of one half workgroup
struct DwqbPlan {
  int tiles, groups, splits, rows_per_split, items, wgs;
};
inline DwqbPlan cin_dwqb_plan(long M, int C, int cus) {
  DwqbPlan p;
  p.tiles = (C + 31) / 32;
  p.groups = (p.tiles + 3) / 4;
  long want = std::max<long>(1, 2L * cus / p.groups);           // one item per workgroup, two workgroups per CU, all resident at once
  want = std::max<long>(want, (M + (1L << 20) - 1) >> 20);      // byte offsets inside a split (rows * 512) stay below 2^31
  const long rows = std::max<long>(16, ((M + want - 1) / want + 15) / 16 * 16);
  p.rows_per_split = (int)rows;
  p.splits = (int)std::max<long>(1, (M + rows - 1) / rows);
  p.items = p.groups * p.splits;
  p.wgs = p.items;
  return p;
}

template <int XS = 3>   // xe ring slots (a template so that only the translation unit that launches it compiles it)
__global__ __launch_bounds__(256, 2) void cin_dwq_b_kernel(const float* __restrict__ gT, const float* __restrict__ x1T, int HS, const float* __restrict__ xe,
                                                           int XE, float* __restrict__ part, int M, int F, int symD, int rows_per_split, int splits,
                                                           int groups, int items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // planes 2 x 24 KB | xe 3 x 4 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, half = lane >> 5;
  const int w4 = wave;
  const int Cp = F * symD, C = Cp + F, tiles = (C + 31) >> 5;
  // XCD-aware work mapping (as cin_dwq_kernel): an XCD streams only its own row splits through its L2
  const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int item = wg;
  if (item >= items) return;                         // (whole workgroup)
  const bool live = true;
  const int split = live ? item / groups : 0, grp = live ? item % groups : 0;
  const int tile = grp * 4 + w4;
  const bool work = live && tile < tiles;            // (a wave without a tile still cuts its share of the B tiles)
  const int c0 = tile * 32;
  const long m_lo = (long)split * rows_per_split;
  const long mrem = live ? std::max<long>(0, std::min<long>(M, m_lo + rows_per_split) - m_lo) : 0;   // rows past it read as zeros
  const __amdgpu_buffer_rsrc_t rg = make_rsrc_uniform(gT + (live ? m_lo * HS : 0), mrem * HS * 4);
  const __amdgpu_buffer_rsrc_t r1 = make_rsrc_uniform(x1T + (live ? m_lo * HS : 0), mrem * HS * 4);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(xe + (live ? m_lo * XE : 0), mrem * XE * 4);
  unsigned char* hbase = lds;
  unsigned char* xring = hbase + 2 * kDwqbPlaneSlot;
  const int steps = rows_per_split >> 4;
  const int c = c0 + r;
  const int cc = c < C ? c : C - 1;
  int hh, ff, sel;
  if (cc < Cp) {
    hh = cc / symD;
    ff = (hh + (cc - hh * symD)) % F;
    sel = 0;
  } else {
    hh = F;
    ff = cc - Cp;
    sel = 1;
  }
  const int so = F + 1 + sel;
  auto dma_xe = [&](int t) {   // this wave's 1-KB piece of the step's 16 rows of xe (a slot is 4 KB: the rows and what follows them)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xring + (t % XS) * kDwqbXeSlot + w4 * 1024), 16, lane * 16,
                                             __builtin_amdgcn_readfirstlane(t * 64 * XE + w4 * 1024), 0, 0);
  };
  // this thread's share of a step's B tile: column 4 r + w4 of G1 and of x1, rows rho(half, 0..7)
  float raw[2][8];
  const int bvo = (4 * half * HS + 4 * r + w4) * 4;
  auto load_raw = [&](int t) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sof = __builtin_amdgcn_readfirstlane((16 * t + (e & 3) + 8 * (e >> 2)) * HS * 4);
      raw[0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, bvo, sof, 0));
      raw[1][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, bvo, sof, 0));
    }
  };
  auto cut_raw = [&](int t) {   // -> the planes of step t, ring slot t % 2
    unsigned char* dst = hbase + (t & 1) * kDwqbPlaneSlot + lane * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 a[3];
      split3(raw[j], a);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(dst + (pl * 8 + j * 4 + w4) * 1024) = a[pl];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  dma_xe(0);
  load_raw(0);
  dma_xe(1);
  cut_raw(0);
  load_raw(1);
#pragma unroll 1
  for (int t = 0; t < steps; ++t) {
    // xe of step t was issued two steps ago; younger operations may stay in flight: the DMA piece and the 16 loads of step t+1
    asm volatile("s_waitcnt vmcnt(17) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // xe(t) and every wave's planes(t) are in LDS; nobody reads xe slot (t+2)%3 or plane slot (t+1)%2 any more
    dma_xe(t + 2);
    cut_raw(t + 1);
    load_raw(t + 2);
    if (work) {
      // A planes of this step from the xe rows in LDS
      const float* xs = reinterpret_cast<const float*>(xring + (t % XS) * kDwqbXeSlot);
      float pa[8], pb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* row = xs + ((e & 3) + 8 * (e >> 2) + 4 * half) * XE;
        pa[e] = row[hh] * row[ff];
        pb[e] = pa[e] * row[so];
      }
      u32x4 a0[3], a1[3];
      split3(pa, a0);
      split3(pb, a1);
      const unsigned char* sb = hbase + (t & 1) * kDwqbPlaneSlot + lane * 16;
      u32x4 b[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[0][pl] = *reinterpret_cast<const u32x4*>(sb + (pl * 8) * 1024);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        if (nb + 1 < 8) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[(nb + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(sb + (pl * 8 + nb + 1) * 1024);
        }
        acc[nb] = nb < 4 ? mfma_split(a0, b[nb & 1], acc[nb]) : mfma_split(a1, b[nb & 1], acc[nb]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (trailing DMA pieces: nothing may land in LDS after the workgroup is gone)
  if (!work) return;
  float* pout = part + (long)split * C * 256;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int cr = c0 + mfma32_row(reg, half);
    if (cr < C) {
      float* dst = pout + (long)cr * 256 + 4 * r;
      if (cr < Cp) *reinterpret_cast<float4*>(dst) = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
      *reinterpret_cast<float4*>(dst + 128) = make_float4(acc[4][reg], acc[5][reg], acc[6][reg], acc[7][reg]);
    }
  }
}

void cin_launch_dwq_b(hipStream_t st, const DwqbPlan& p, const float* gT, const float* x1T, int HS, const float* xe, int XE, float* part, int M, int F,
                      int symD);

bool cin_launch_fwdq_b(hipStream_t st, int JT, const float* x2T, int XL, const u32x4* Wb, int NT, const float* bias1, const float* wsn, int JTG,
                       const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf);

}  // namespace fil
