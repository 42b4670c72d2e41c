// Quadratic tail of a three-layer CIN, merged weight gradients (round 4): ONE weight-gradient GEMM for the first layer and the quadratic
// form together instead of two launches (cin_qtail.h explains the algebra; reference interactive_layer.py:310-327).
//
// With P[m,c] = x[m,h_c] x[m,f_c] over the unordered field pairs c (cin_pack_wf_sym_kernel's pair weights W1s, Ts):
//   x1 = P W1s + b1,  R = P Ts,  pool_L = <x1, R> + ...                                   (forward: cin_fwd3_kernel, twice)
//   [dW1s | dTs] = P^T [G1 | dP_L x1]                                                       -> cin_dwq_kernel: 256 output columns
// The generated operand (two gathers and a multiply per step) is paid once per 256 columns instead of once per 128, the channel
// tiles are dealt so that every wave of the grid carries the same number of steps (the 128-column launches left 13 % of their wave
// slots idle: 780 pair rows are 24.4 tiles of 32), and one reduction pass instead of two sums the row-split partials.
// (The matching merge of the two data-gradient launches -- reduction length 256, [G1 | dP_L x1] [W1s ; Ts]^T -- was built and dropped:
// the lane's two half rows then take 128 registers per 32 rows, which forces one wave per SIMD, and with 64 rows per wave the
// compiler parks the B operands in accumulator registers and copies one back in front of every MFMA.)
#pragma once
#include "cin_kernels.h"
#include "cin_launch.h"
#include "cin_qtail.h"

namespace fil {

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradients.  Wave = 32 channel rows (A-operand rows, on the lanes) x 256 columns [G1 | x1]; the reduction runs over the rows
// m of one row split, two per step (one per wave half), every operand streamed through a DEPTH-step register queue by raw buffer
// loads with scalar step offsets (as cin_dw3_kernel):
//   channel c < Cp = F*symD: the unordered pair (h, f = (h + d) mod F), c = h*symD + d;   c in [Cp, Cp + F): field f = c - Cp alone
//   xe [M][XE = F+3] = x[m,0..F-1] | 1 | dP_L[m] | dP_p[m]           (cin_qtail_xe_kernel)
//   A [c][m] = xe[m,h_c] xe[m,f_c]  (h_c = F, the ones column, for the single-field rows)      -> columns   0..127 (x G1):  dW1 pairs
//   A'[c][m] = A[c][m] * (c < Cp ? dP_L[m] : dP_p[m])                                          -> columns 128..255 (x x1):  dT pairs | v^T
// i.e. the scale of the quadratic form sits on the generated operand (one more gathered dword and multiply per step), nothing of
// size [M, 128] is written for it.  Work: a workgroup = 8 waves = 4 channel tiles x the two row splits of a pair (the four waves of
// a split share its B rows through L1; the two waves of a tile fold their accumulators through LDS into ONE partial); the tiles left
// over after the last full group of four (26 tiles at F = 39: 2) form workgroups of their own in which the spare waves take further
// pairs, so every wave of the grid carries the same number of steps.
constexpr int kDwqDepth = 6;
constexpr int kDwqThreads = 512;   // 8 waves: 4 channel tiles x 2 row splits
struct DwqPlan {
  int tiles, ncol_full, rem, splits, pairs, rows_per_split, wgs_full, wgs;
};
// leftover workgroups for `pairs` pairs of row splits: rem = 1 tile: 4 pairs per workgroup, 2 tiles: 2 pairs, 3 tiles: 1 pair (2 idle waves)
inline int cin_dwq_rem_wgs(int rem, int pairs) { return rem == 0 ? 0 : (rem == 1 ? (pairs + 3) / 4 : (rem == 2 ? (pairs + 1) / 2 : pairs)); }
inline DwqPlan cin_dwq_plan(long M, int C, int cus) {
  DwqPlan p;
  p.tiles = (C + 31) / 32;
  p.ncol_full = p.tiles / 4;
  p.rem = p.tiles % 4;
  const long unit = 2 * kDwqDepth;
  const long slots = cus;                                // one 8-wave workgroup (two waves per SIMD) per CU, all resident at once
  int best = 1;
  for (int s = 1; s <= 512; ++s) {
    if ((long)p.ncol_full * s + cin_dwq_rem_wgs(p.rem, s) > slots) break;
    best = s;
  }
  long want = std::max<long>(2L * best, (M + (1L << 20) - 1) >> 20);    // byte offsets inside a split (rows * 1 KiB) stay below 2^31
  long rows = std::max(unit, ((M + want - 1) / want + unit - 1) / unit * unit);
  p.rows_per_split = (int)rows;
  p.splits = (int)std::max<long>(1, (M + rows - 1) / rows);
  p.pairs = (p.splits + 1) / 2;
  p.wgs_full = p.ncol_full * p.pairs;
  p.wgs = p.wgs_full + cin_dwq_rem_wgs(p.rem, p.pairs);
  return p;
}

// FOLD = 4: a workgroup = 2 channel tiles x the FOUR row splits of a quad; the quad's accumulators meet through LDS in two rounds --
// (s0 + s1) + (s2 + s3) -- and ONE partial leaves per quad: half the partial-sum traffic again (written here, read by the reduce launch).
// Two waves instead of four share a split's B rows through L1.  Leftover tile (odd tile count): workgroups of their own, two quads each.
inline DwqPlan cin_dwq_plan4(long M, int C, int cus) {
  DwqPlan p;
  p.tiles = (C + 31) / 32;
  p.ncol_full = p.tiles / 2;   // tile pairs
  p.rem = p.tiles % 2;
  const long unit = 2 * kDwqDepth;
  int best = 1;                // quads
  for (int q = 1; q <= 256; ++q) {
    if ((long)p.ncol_full * q + (p.rem ? (q + 1) / 2 : 0) > cus) break;
    best = q;
  }
  long want = std::max<long>(4L * best, (M + (1L << 20) - 1) >> 20);
  long rows = std::max(unit, ((M + want - 1) / want + unit - 1) / unit * unit);
  p.rows_per_split = (int)rows;
  p.splits = (int)std::max<long>(1, (M + rows - 1) / rows);
  p.pairs = (p.splits + 3) / 4;   // (quads: the partials this launch leaves)
  p.wgs_full = p.ncol_full * p.pairs;
  p.wgs = p.wgs_full + (p.rem ? (p.pairs + 1) / 2 : 0);
  return p;
}

template <int DEPTH = kDwqDepth, int FOLD = 2>
__global__ __launch_bounds__(kDwqThreads, 2) void cin_dwq_kernel(const float* __restrict__ gT, const float* __restrict__ x1T, int HS, const float* __restrict__ xe,
                                                                int XE, float* __restrict__ part, int M, int F, int symD, int rows_per_split, int splits,
                                                                int ncol_full, int rem, int wgs_full, int wgs) {
  // waves w and w + 4 of a workgroup take the same channel tile and the two row splits of one PAIR; the upper one hands its
  // accumulators over through LDS and the lower one stores the sum: one partial per pair of splits (half the partial-sum traffic of
  // the reduction pass, in a fixed order: lower split + upper split)
  __shared__ float fold[4][128][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int Cp = F * symD, C = Cp + F;
  // XCD-aware work mapping (as cin_dw3_kernel): workgroup i of XCD i%8 takes item (i%8)*(grid/8) + i/8 of the split-major list
  const int item = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (item >= wgs) return;   // (whole workgroup)
  const int wl = FOLD == 4 ? (wave & 1) : (wave & 3), up = FOLD == 4 ? (wave >> 1) : (wave >> 2);
  int tile, pair;
  bool idle = false;
  if constexpr (FOLD == 4) {   // (pair = the quad of row splits)
    if (item < wgs_full) {
      tile = (item % ncol_full) * 2 + wl;
      pair = item / ncol_full;
    } else {
      tile = ncol_full * 2;
      pair = 2 * (item - wgs_full) + wl;
    }
  } else if (item < wgs_full) {
    tile = (item % ncol_full) * 4 + wl;
    pair = item / ncol_full;
  } else {
    const int j = item - wgs_full;
    if (rem == 1) {
      tile = ncol_full * 4;
      pair = 4 * j + wl;
    } else if (rem == 2) {
      tile = ncol_full * 4 + (wl & 1);
      pair = 2 * j + (wl >> 1);
    } else {
      tile = ncol_full * 4 + wl;
      pair = j;
      idle = wl == 3;
    }
  }
  tile = __builtin_amdgcn_readfirstlane(tile);
  pair = __builtin_amdgcn_readfirstlane(pair);
  const int split = FOLD * pair + up;
  const int npairs = (splits + FOLD - 1) / FOLD;
  const int c0 = tile * 32;
  const int m_lo = split * rows_per_split;
  const int m_hi = min(M, m_lo + rows_per_split);
  const bool work = !idle && pair < npairs && split < splits && m_lo < m_hi;   // (a wave without rows still takes part in the hand-over)
  // the descriptors cover exactly the split's rows: the last step's spare row and the prefetch past the end read zeros
  const long mrem = work ? (long)m_hi - m_lo : 0;
  const long mbase = work ? m_lo : 0;
  const __amdgpu_buffer_rsrc_t rg = make_rsrc_uniform(gT + mbase * HS, mrem * HS * 4);
  const __amdgpu_buffer_rsrc_t r1 = make_rsrc_uniform(x1T + mbase * HS, mrem * HS * 4);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(xe + mbase * XE, mrem * XE * 4);
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
  const int ho = (half * XE + hh) * 4, fo = (half * XE + ff) * 4, so = (half * XE + F + 1 + sel) * 4;
  const int go = (half * HS + 4 * r) * 4;
  const int steps = work ? (m_hi - m_lo + 1) >> 1 : 0;
  const int groups = (steps + DEPTH - 1) / DEPTH;

  f32x16 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  f32x4s qg[DEPTH], q1[DEPTH];
  float qh[DEPTH], qf[DEPTH], qs[DEPTH];
  auto fetch = [&](int s, int d) {
    const int row = 2 * s;   // uniform, relative to the split's first row
    qg[d] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rg, go, row * HS * 4, 0));
    q1[d] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(r1, go, row * HS * 4, 0));
    qh[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ho, row * XE * 4, 0));
    qf[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, fo, row * XE * 4, 0));
    qs[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, so, row * XE * 4, 0));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, d);
  // the generated operands of step s+1 are computed in front of step s's MFMAs and consumed a step later
  float ac = qh[0] * qf[0], ac2 = ac * qs[0], an, an2;
#pragma unroll 1
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const f32x4s g4 = qg[d], x4 = q1[d];
      an = qh[(d + 1) % DEPTH] * qf[(d + 1) % DEPTH];   // (slot 0 of the next group was refilled a group ago)
      an2 = an * qs[(d + 1) % DEPTH];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma32(ac, g4[0], acc[0]);
      acc[1] = mfma32(ac, g4[1], acc[1]);
      acc[2] = mfma32(ac, g4[2], acc[2]);
      acc[3] = mfma32(ac, g4[3], acc[3]);
      acc[4] = mfma32(ac2, x4[0], acc[4]);
      acc[5] = mfma32(ac2, x4[1], acc[5]);
      acc[6] = mfma32(ac2, x4[2], acc[6]);
      acc[7] = mfma32(ac2, x4[3], acc[7]);
      ac = an;
      ac2 = an2;
      __builtin_amdgcn_sched_barrier(0);
      fetch(g * DEPTH + d + DEPTH, d);   // after the step's MFMAs: the refill may land in the registers it replaces
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // hand-over: the upper wave's 128 accumulator registers through LDS ([register][lane]: conflict-free), then lower + upper
  if constexpr (FOLD == 4) {
    // round 1: split 1 -> split 0 and split 3 -> split 2 (four slots); round 2: split 2 -> split 0
    const int slot = (up >> 1) * 2 + wl;
    if (up & 1) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) fold[slot][nb * 16 + i][lane] = acc[nb][i];
    }
    __syncthreads();
    if (!(up & 1)) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] += fold[slot][nb * 16 + i][lane];
    }
    __syncthreads();
    if (up == 2) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) fold[wl][nb * 16 + i][lane] = acc[nb][i];
    }
    __syncthreads();
    if (up != 0 || pair >= npairs) return;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] += fold[wl][nb * 16 + i][lane];
  } else {
    if (up == 1) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) fold[wl][nb * 16 + i][lane] = acc[nb][i];
    }
    __syncthreads();
    if (up == 1 || idle || pair >= npairs) return;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] += fold[wl][nb * 16 + i][lane];
  }
  float* pout = part + (long)pair * C * 256;
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

// Fixed-order sum of the partials [pairs of row splits][C][256] of cin_dwq_kernel (64 outputs per workgroup, the 4 waves take every
// 4th partial, as cin_reduce_kernel), written straight to their destinations:
//   pair row c = h*D + d, column n < H1:        dW1[(h,f), n] and, unless d == 0 or 2d == F, dW1[(f,h), n]      (f = (h + d) mod F)
//   pair row c, column 128 + n, n < H2:         dT[(h,f), n] / dT[(f,h), n] likewise  ([F*F][H2])
//   single-field row c = Cp + f, column 128+n:  vT[f][n]                                                          ([F][H2])
// Workgroups past those rows: the fixed-order sum of the nbp column-sum partials bpart [nbp][H1] -> dbias1 (cin_reduce_body).
static __global__ __launch_bounds__(256) void cin_reduce_expand_q_kernel(const float* __restrict__ part, int parts, int F, int D, int H1, int H2,
                                                                          float* __restrict__ dW1, float* __restrict__ dT, float* __restrict__ vT,
                                                                          const float* __restrict__ bpart, int nbp, float* __restrict__ dbias1,
                                                                          int nb1, const float* __restrict__ hpart, int nhp, int LK,
                                                                          float* __restrict__ ddw, float* __restrict__ ddb) {
  const int Cp = F * D, C = Cp + F;
  const int nbw = (C + 3) / 4;   // a wave per row c (all 256 columns, four per lane), four rows per workgroup
  if ((int)blockIdx.x >= nbw + nb1) {   // the dense head's block partials (cin_qtail_xe_kernel) -> ddense_w | ddense_b
    cin_reduce_body(hpart, ddw, (long)LK + 1, nhp, ddb, (long)LK, (int)blockIdx.x - nbw - nb1);
    return;
  }
  if ((int)blockIdx.x >= nbw) {
    cin_reduce_body(bpart, dbias1, (long)H1, nbp, nullptr, 0, (int)blockIdx.x - nbw);
    return;
  }
  // (round 5: 16-byte loads, eight partials in flight per lane, no LDS and no barrier -- 205 workgroups instead of 3276: 16.7 -> 13.3 us)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wave;
  if (c >= C) return;
  const long ps = (long)C * 256;
  const float* p0 = part + (long)c * 256 + 4 * lane;
  f32x4 t[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int p = 0;
  for (; p + 7 < parts; p += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += *reinterpret_cast<const f32x4*>(p0 + (long)(p + u) * ps);
  }
  for (; p < parts; ++p) t[0] += *reinterpret_cast<const f32x4*>(p0 + (long)p * ps);
  const f32x4 v = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  const int col = 4 * lane, n0 = col & 127;
  const bool second = col >= 128;
  const int H = second ? H2 : H1;
  if (c >= Cp) {
    if (second) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + e < H2) vT[(long)(c - Cp) * H2 + n0 + e] = v[e];
    }
    return;
  }
  const int h = c / D, d = c - h * D;
  const int f = (h + d) % F;
  float* dst = second ? dT : dW1;
  const bool twin = d != 0 && 2 * d != F;
  float* r0 = dst + ((long)h * F + f) * H + n0;
  float* r1 = dst + ((long)f * F + h) * H + n0;
  if ((H & 3) == 0 && n0 + 3 < H) {
    *reinterpret_cast<f32x4*>(r0) = v;
    if (twin) *reinterpret_cast<f32x4*>(r1) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n0 + e < H) {
        r0[e] = v[e];
        if (twin) r1[e] = v[e];
      }
  }
}

// xe[m][XE = F+3] = x[m,0..F-1] | 1 | dP_L[m] | dP_p[m]  (operand rows of cin_dwq_kernel), and per block of 256 rows the column sums
// dcpart[blk][f] = sum_m dP_L[m] x[m,f] (-> dc[f] = d pool_L / d c[f]), dcpart[blk][F] = sum_m dP_L[m], dcpart[blk][F+1] = sum_m dP_p[m]
// (cf. cin_qtail_scale_kernel).  Further workgroups: [nscale, +nhead) the dense head's partial sums -> ddense_w | ddense_b; the
// rest: the first layer's pair weights in the dZ kernel's slot order.   LDS: [256][F + 3 + L].  kXeThreads threads.
constexpr int kXeThreads = 1024;
static __global__ __launch_bounds__(kXeThreads) void cin_qtail_xe_kernel(const float* __restrict__ xT, const float* __restrict__ dPL,
                                                                  const float* __restrict__ dPp, int ldp, int K, float* __restrict__ xe,
                                                                  float* __restrict__ dcpart, int M, int F, int nscale,
                                                                  const float* __restrict__ g, const float* __restrict__ dense_w,
                                                                  const float* __restrict__ pooled, float* __restrict__ dP, float* __restrict__ hpart,
                                                                  int LK, int lL, int lp, const float* __restrict__ W0, float* __restrict__ Wz,
                                                                  int H0, int JTs, int HS0, int tiles0, int npz = -1,
                                                                  const float* __restrict__ Tq = nullptr, int HT = 0, u32x4* __restrict__ Wzb1 = nullptr,
                                                                  u32x4* __restrict__ Wzb2 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= nscale) {   // the first layer's weights in the dZ kernel's slot order (nothing else uses that buffer here)
    if (threadIdx.x >= 256) return;   // (the pack bodies are written for 256 threads)
    if (npz < 0) npz = gridDim.x - nscale;
    const int b = blockIdx.x - nscale;
    if (b < npz) cin_pack_wz_sym_body(W0, Wz, F, H0, JTs, HS0, tiles0, b, npz);
    else {   // split-bf16 mode: both layers' slot-ordered weights as planes (cin_qs_pack_wz_body), half of the extra workgroups each
      const int nq = ((int)gridDim.x - nscale - npz) >> 1, q = b - npz;
      if (q < nq) cin_qs_pack_wz_body(W0, H0, Wzb1, tiles0, F, JTs, q, nq);
      else if (q < 2 * nq) cin_qs_pack_wz_body(Tq, HT, Wzb2, tiles0, F, JTs, q - nq, nq);
    }
    return;
  }
  // LDS: xs [256 F] (the block's rows of xT as they lie in memory) | dls [256] | dps [256] | hd [256][nl + 1] (head: g * pooled of the
  // row's column in each of the L = LK/K layers | g on the sample's first row).  The block's rows of xT and of xe are each one
  // contiguous range: both move by coalesced (16-byte) accesses through the image (one row per thread straight from memory was
  // F + XE wave instructions with 64 lanes in 64 different lines each).  1024 threads: all of them fetch, then twelve waves write
  // the xe rows while the other four take the column sums and the head's partials.
  const int nl = g != nullptr ? LK / K : 0;
  const int hl = nl + 1, XE = F + 3;
  const int tid = threadIdx.x;
  const long m0 = (long)blockIdx.x * 256;
  const int nrows = (int)min((long)256, (long)M - m0);
  float* xs = smem;
  float* dls = xs + 256 * F;
  float* dps = dls + 256;
  float* hd = dps + 256;
  {
    const float* src = xT + m0 * F;
    const int n = nrows * F;
    if ((reinterpret_cast<uintptr_t>(xT) & 15) == 0) {
      const int n4 = n >> 2;
      for (int i0 = tid; i0 < n4; i0 += 4 * kXeThreads) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u * kXeThreads < n4 ? reinterpret_cast<const float4*>(src)[i0 + u * kXeThreads] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u * kXeThreads < n4) reinterpret_cast<float4*>(xs)[i0 + u * kXeThreads] = v[u];
      }
      for (int i = 4 * n4 + tid; i < n; i += kXeThreads) xs[i] = src[i];
    } else {
      for (int i = tid; i < n; i += kXeThreads) xs[i] = src[i];
    }
    for (int i = n + tid; i < 256 * F; i += kXeThreads) xs[i] = 0.f;
  }
  if (tid < 256) {   // one row per thread: its pooled gradients (and the head's terms)
    const long m = m0 + tid;
    float dl = 0.f, dp = 0.f;
    if (m < M) {
      const long b = m / K;
      const int k = (int)(m - b * K);
      if (g != nullptr) {   // the dense head's backward rides here: dP[b, j] = g[b] dense_w[j], and this row's terms of ddense_w | ddense_b
        const float gb = g[b];
        for (int l = 0; l < nl; ++l) {
          const float d = gb * dense_w[l * K + k];
          dP[b * LK + l * K + k] = d;
          hd[tid * hl + l] = gb * pooled[b * LK + l * K + k];
        }
        hd[tid * hl + nl] = k == 0 ? gb : 0.f;
        dl = gb * dense_w[lL * K + k];
        dp = gb * dense_w[lp * K + k];
      } else {
        dl = dPL[b * ldp + k];
        dp = dPp[b * ldp + k];
      }
    } else if (g != nullptr) {
      for (int l = 0; l <= nl; ++l) hd[tid * hl + l] = 0.f;
    }
    dls[tid] = dl;
    dps[tid] = dp;
  }
  __syncthreads();
  __shared__ float cs[4][64];
  if (tid < kXeThreads - 256) {
    // xe rows of the block: x | 1 | dP_L | dP_p, written as the contiguous range they are
    constexpr int NT = kXeThreads - 256;
    float* dst = xe + m0 * XE;
    const int n4 = nrows * XE >> 2;
    const float rxe = 1.f / (float)XE;
    for (int i4 = tid; i4 < n4; i4 += NT) {
      int rr = (int)(((float)(4 * i4) + 0.5f) * rxe);   // (4 i4 < 2^16: the quotient is exact)
      int c = 4 * i4 - rr * XE;
      float e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        e[u] = c < F ? xs[rr * F + c] : c == F ? 1.f : c == F + 1 ? dls[rr] : dps[rr];
        if (++c == XE) {
          c = 0;
          ++rr;
        }
      }
      reinterpret_cast<float4*>(dst)[i4] = make_float4(e[0], e[1], e[2], e[3]);
    }
    for (int i = 4 * n4 + tid; i < nrows * XE; i += NT) {
      const int rr = i / XE, c = i - rr * XE;
      dst[i] = c < F ? xs[rr * F + c] : c == F ? 1.f : c == F + 1 ? dls[rr] : dps[rr];
    }
  } else {
    const int t2 = tid - (kXeThreads - 256);
    // column sums: wave q takes rows 64q .. 64q+63 of column f = lane, the four partial sums meet in wave order
    {
      const int f = t2 & 63, q = t2 >> 6;
      float t0 = 0.f, t1 = 0.f;
      if (f < F) {
        for (int rr = 64 * q; rr < 64 * q + 64; rr += 2) {
          t0 += xs[rr * F + f] * dls[rr];
          t1 += xs[(rr + 1) * F + f] * dls[rr + 1];
        }
      } else if (f < F + 2) {
        const float* dv = f == F ? dls : dps;
        for (int rr = 64 * q; rr < 64 * q + 64; rr += 2) {
          t0 += dv[rr];
          t1 += dv[rr + 1];
        }
      }
      cs[q][f] = t0 + t1;
    }
    // the head's partials of this block: column j = l K + k collects the rows whose k matches, in row order (fixed order); column LK = sum of g
    if (g != nullptr) {
      const int k0 = (int)(m0 % K);
      for (int j = t2; j <= LK; j += 256) {
        float t = 0.f;
        if (j < LK) {
          const int l = j / K, k = j - l * K;
          for (int rr = (k - k0 + K) % K; rr < 256; rr += K) t += hd[rr * hl + l];
        } else {
          for (int rr = (K - k0) % K; rr < 256; rr += K) t += hd[rr * hl + nl];
        }
        hpart[(long)blockIdx.x * (LK + 1) + j] = t;
      }
    }
  }
  __syncthreads();
  if (tid < F + 2) dcpart[(long)blockIdx.x * kQtConst + tid] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Data gradients of the first layer and of the quadratic form in ONE launch, as two passes of one wave over its 32 rows:
//   pass 0:  dZ^T = W1s G1^T           pass 1:  dZ^T = Ts (dP_L x1)^T        (the lane's half row scaled by dP_L[m] on load)
// both contracted into the SAME dX image in LDS (cin_dz3_kernel<.., SYM> explains the slot machinery; this is its exact pair-symmetric
// 32-row form, two waves per SIMD, stripped of the other modes).  Against two launches of that kernel: one launch, one staging of x,
// one pass over dxT, no separate [M][F] arrays for the quadratic form's two gradient halves (nothing is left for the final transpose
// to scale and join), and the Gx term -- sum_f dZ x[m,f], which is dX[m,h] for a pair-symmetric layer -- is added to the dX image
// when its h completes instead of travelling through its own [M][F] array.  The second wave of a SIMD is no longer in lockstep with
// the first after the pass boundary, so each covers the other's operand reload.
//   x entry and dX accumulator of a field sit side by side in LDS, [f][row 128][x | dX]: one 8-byte read per slot, a compare and a select
//   for the wrap of f, everything else compile-time offsets of the LDS instructions; the weights stream through a 16-deep register queue
//   by scalar-offset buffer loads; x[m,h] of a period comes from the LDS image.
constexpr int kDz2FieldStride = 128 * 2;   // floats per field: 128 rows x (x, dX)
inline int cin_dz2_rows(int F, int JT) {   // LDS field rows: the wrapped slot fields (cin_dz_sym_rows) and the x entries of every period's h values
  const int hpp = cin_dz_h_per_period(JT);
  return std::max(cin_dz_sym_rows(F, JT), (F + hpp - 1) / hpp * hpp);
}

// What cin_dz2_kernel (and its split-bf16 form, cin_qsplit.h) does with the finished dX image of a wave's 32 rows.
__device__ __forceinline__ void cin_dz2_finish(const float* smem, int wave, int lane, int wrow0, int M, int F, int K, const float* __restrict__ dsc, int ldp,
                                               float* __restrict__ dxT, int accumulate, float* __restrict__ dx, const float* __restrict__ cvec) {
  constexpr int FS = kDz2FieldStride;
  const float* img = smem + wave * 32 * 2 + 1;
  const int nrow = min(32, M - wrow0);
  if (dx != nullptr) {
    // final form: dx [B,F,K] = (what dxT holds: the shortcut's part) + the image + dP_L[m] c[f], transposed on the way out -- no
    // separate transpose launch.  Lanes walk the wave's rows fastest: consecutive k of a sample are consecutive floats of dx.
    const int rr = lane & 31;
    const int mm = wrow0 + rr;
    const bool ok = rr < nrow;
    const long mc = ok ? mm : M - 1;
    const long bb = mc / K;
    const int kk = (int)(mc - bb * K);
    const float sc = dsc[bb * ldp + kk];
    float* dxb = dx + (bb * F) * K + kk;
    const float* add = dxT + mc * F;
    // (eight loads in flight per batch: one at a time they were F/2 exposed memory latencies in a row at the very end of the wave)
    for (int f0 = lane >> 5; f0 < F; f0 += 16) {
      float a8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] = accumulate ? add[min(f0 + 2 * u, F - 1)] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + 2 * u;
        if (f < F) {
          const float v = a8[u] + img[f * FS + rr * 2] + sc * cvec[f];
          if (ok) dxb[(long)f * K] = v;
        }
      }
    }
    return;
  }
  // dX rows of the wave are contiguous in dxT ([32 rows][F]): written cooperatively from the LDS image, whole lines per store
  float* dst = dxT + (long)wrow0 * F;
  for (int idx = lane; idx < nrow * F; idx += 64) {
    const int rr = idx / F, f = idx - rr * F;
    const float v = img[f * FS + rr * 2];
    dst[idx] = accumulate ? dst[idx] + v : v;
  }
}

template <int JT, int G>
__global__ __launch_bounds__(256, 2) void cin_dz2_kernel(const float* __restrict__ g1T, const float* __restrict__ g2T, int HS,
                                                         const float* __restrict__ dsc, int ldp, int K, const float* __restrict__ Wz1,
                                                         const float* __restrict__ Wz2, const float* __restrict__ xT, float* __restrict__ dxT,
                                                         int accumulate, int M, int F, int H1, int H2, int periods, int FR,
                                                         float* __restrict__ dx, const float* __restrict__ cvec) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [FR][128 rows][2]
  constexpr int P = JT / gcd_c(16, JT);
  constexpr int HPP = 16 * P / JT;
  constexpr int NQ = 16;                               // float4 per tile and lane (64 columns per wave half)
  constexpr int FS = kDz2FieldStride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M) return;   // (no workgroup barriers in this kernel)
  float* lrow = smem + (wave * 32 + r) * 2;            // this lane's row: field f at lrow[f*FS + {0: x, 1: dX}]
  const int m = wrow0 + r;
  const bool vq = m < M;
  const long mq = vq ? m : M - 1;
  // the lane's half row of the first pass: 16-byte loads, issued before everything else (in flight behind the staging of x)
  f32x4s gq[16];
  {
    const f32x4s* grow4 = reinterpret_cast<const f32x4s*>(g1T + mq * HS + half * 64);
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) gq[s4] = grow4[s4];
  }
  // x entries (zero past F and for rows past M) and zeroed dX accumulators: eight loads per batch, then the LDS writes
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
  __builtin_amdgcn_wave_barrier();  // the halves of a row read each other's x entries from here on
  const long wbytes = ((long)periods * P + 1) * 32 * 128 * 4;
  const int wo = (r * 128 + half * 64) * 4;
  // B operand of the current pass: the lane's half row, columns past the layer's width and rows past M zeroed, scaled (pass 1: by dP_L)
  float greg[64];
  auto to_greg = [&](const f32x4s (&gv4)[16], int Hk, float sc) {
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int keep = (vq && half * 64 + 4 * s4 + e < Hk) ? -1 : 0;
        const float gv = gv4[s4][e];   // (a copy: __builtin_bit_cast applied to the vector ELEMENT expression reads element 0 for every e)
        greg[4 * s4 + e] = __builtin_bit_cast(float, __builtin_bit_cast(int, gv) & keep) * sc;
      }
  };
  to_greg(gq, H1, 1.f);

#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {   // (running the passes in the other order in half of the workgroups, so that the two waves of a
    // SIMD do not reload their operands at the same moment, changed nothing; neither did a real stagger -- every other group of eight
    // workgroups walking pass 0 [periods/2, end) | pass 1 | pass 0 [0, periods/2), their reloads in the middle of the others' passes:
    // 225.6 -> 227-229 us, the third reload costs more than the spread returns)
    const float* gT = pass == 0 ? g1T : g2T;
    const int Hk = pass == 0 ? H1 : H2;
    const float sc = pass == 0 ? 1.f : dpl;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(pass == 0 ? Wz1 : Wz2, wbytes);
    auto ldw = [&](int t, int s4) {   // tile t is [32 slot rows][128]; lane (r, half) reads row r, columns half*64 + 4*s4 .. +3
      return __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rw, wo + 16 * s4, t * (32 * 128 * 4), 0));
    };
    // queue depth in step groups (half a tile ahead: 8 x 256 cycles covers the L2 latency; the full tile, 16, buys nothing: 0.240 / 0.244 ms
    // against 0.240 / 0.237 on one box, at 255 registers instead of 218)
    constexpr int QD = 8;
    f32x4s q[QD];
#pragma unroll
    for (int s4 = 0; s4 < QD; ++s4) q[s4] = ldw(0, s4);
    // the lane's half row of the second pass (the first pass's was fetched at the top, behind the staging of x)
    if (pass == 1) {
      const f32x4s* grow4 = reinterpret_cast<const f32x4s*>(gT + mq * HS + half * 64);
      f32x4s g2[16];
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) g2[s4] = grow4[s4];
      to_greg(g2, Hk, sc);
    }
    float gx = 0.f;
    f32x16 dprev;
#pragma unroll
    for (int i = 0; i < 16; ++i) dprev[i] = 0.f;
    float xprev[HPP], xcur[HPP];
#pragma unroll
    for (int hl = 0; hl < HPP; ++hl) xprev[hl] = xcur[hl] = 0.f;
    int hprev = 0;   // h base of the period the previous tile belongs to (the fake tile before the first one: dZ = 0, any valid rows)
    // A slot is contracted in two halves one step group apart (read, then FMA + write a group later): the LDS latency stays off the
    // MFMA chain.  Within a group the apply (write) precedes the next fetch (read), so slots that alias one word stay ordered.
    // Slots per block: G slots are fetched together behind a block of 4 G MFMAs and applied behind the next block, the blocks pinned
    // by scheduling barriers.  One slot per step group (G = 1, pinned: cin_dz3_kernel's form) 0.280 ms; G = 1 left to the compiler
    // 0.248; G = 2 / 4 / 8 unpinned 0.242 / 0.235 / 0.235; G = 4 pinned 0.2285; G = 8 pinned 0.234: runs of 16 dependent MFMAs
    // keep the pipe fed by the SIMD's other wave while this one does its four LDS round trips in one go.
    // G > 1 needs the G slots of a block (and the rows completed while they are pending) to be distinct words: slots alias at a
    // distance of 2 JT - 1 (offsets hl + 2j) or through the wrap of f -- the launcher picks G = 4 for JT >= 4 and
    // F >= HPP + 2 JT (no two offsets of a period differ by F), G = 1 otherwise.
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
      const int off = sp / JT + 2 * (sp % JT);   // compile-time after unrolling: f = (h + off + half) mod F, h + off + half < F + FR
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
        // h = hb + hl is complete: dX[m,h] += sum over both lane halves (one LDS add by the lower half; the row's own words only)
        const float t = lane_halves_sum(gx);
        gx = 0.f;
        if (half == 0) __hip_atomic_fetch_add(lrow + (hb + hl) * FS + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    };
    sym_period(hprev);
#pragma unroll
    for (int k = 0; k < G; ++k) slot_fetch(P - 1, k, k);
#pragma unroll 1
    for (int per = 0; per < periods; ++per) {
      const int hbase = per * HPP;
      // x[m, hbase + hl] from the LDS image (rows >= F hold zeros; FR >= periods * HPP)
#pragma unroll
      for (int hl = 0; hl < HPP; ++hl) xcur[hl] = lrow[(hbase + hl) * FS];
#pragma unroll
      for (int tp = 0; tp < P; ++tp) {
        const int t = per * P + tp;
        f32x16 d;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < NQ; ++s4) {
          const f32x4s w = q[s4 % QD];
          d = mfma32(w[0], greg[4 * s4 + 0], d);
          d = mfma32(w[1], greg[4 * s4 + 1], d);
          d = mfma32(w[2], greg[4 * s4 + 2], d);
          d = mfma32(w[3], greg[4 * s4 + 3], d);
          q[s4 % QD] = s4 + QD < NQ ? ldw(t, s4 + QD) : ldw(t + 1, s4 + QD - NQ);   // (the stream is allocated one tile past the last period)
          if (s4 % G == G - 1) {
            // the previous tile's slots s4-G+1 .. s4: applied (their words were fetched a block ago), then the next G fetched.  An
            // aliasing word is written before it is fetched again: applies precede fetches, and the slots of one block are distinct.
#pragma unroll
            for (int k = 0; k < G; ++k) {
              const int rr = s4 - G + 1 + k;
              if (tp == 0) slot_apply(dprev, xprev, hprev, P - 1, rr, k);
              else slot_apply(dprev, xcur, hbase, tp - 1, rr, k);
            }
            if (s4 < 15) {
#pragma unroll
              for (int k = 0; k < G; ++k) slot_fetch(tp == 0 ? P - 1 : tp - 1, s4 + 1 + k, k);
            } else {
              if (tp == 0) sym_period(hbase);   // (from here on the slots belong to this period)
#pragma unroll
              for (int k = 0; k < G; ++k) slot_fetch(tp, k, k);   // first slots of this tile, applied behind the first block of the next one
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
    // the last tile's slots
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

// dx != nullptr: the kernel finishes the job -- dx [B,F,K] = transpose(dxT (read only) + its image) + dP_L c -- instead of updating dxT
bool cin_launch_dz2(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1,
                    const float* Wz2, const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods,
                    float* dx = nullptr, const float* cvec = nullptr);

// ------------------------------------------------------------------------------------------------------------------------------
// Forward of the first layer and of the quadratic form in ONE launch: [x1 | R] = pairs(x) [W1s | Ts] (+ b1), 256 output columns.
// The sum-pool relayout and Dense(1) head folded into cin_fwdq_kernel's epilogue (three layers in the order first | p | L, K a power of
// two <= 32: a wave's 32 rows are whole samples).  pooled == nullptr: the kernel leaves per-layer pool arrays for cin_head_fwd_kernel.
struct CinHeadFold {
  float* pooled;          // [B][LK]
  float* out;             // [B] or nullptr (output_dim != 1)
  const float* dense_w;   // [LK]
  const float* dense_b;
  int kshift, LK, op, oL; // K = 1 << kshift; column offsets p K and (L-1) K of the two upper layers
};

// Everything of the merged forward behind its main loop (shared by the exact kernel below and the split-bf16 kernel of cin_qsplit.h,
// which differ only in how acc -- [x1 | R] of the wave's 32 rows, lane r owning columns 4r..4r+3 of each half -- was accumulated):
// S = x wsum_p^T and <x, c> from the wrapped rows (exact fp32 MFMA), + b1, the stores of x1 and R, the three sum-pools and, with the
// head folded in, pooled [B, L K] and the Dense(1) output.  NW = waves per workgroup (rows of the two small LDS arrays).
template <int NW>
__device__ __forceinline__ void cin_fwdq_epilogue(f32x16 (&acc)[8], const __amdgpu_buffer_rsrc_t& rx, int vhalf, int wo, int r, int half, int wave, int wrow0,
                                                  const float* __restrict__ bias1, const float* __restrict__ wsn, int JTG, const float* __restrict__ cvec,
                                                  float* __restrict__ x1T, float* __restrict__ RT, int HS, float* __restrict__ pool1,
                                                  float* __restrict__ pool_p, float* __restrict__ pool_L, int M, int F, int H, const CinHeadFold& hf,
                                                  float (&lin_s)[NW][32], float (&pv_s)[NW][3][32]) {
  const int lane = half * 32 + r;
  // ---- S = x wsum_p^T and the linear term <x, c>: JTG steps of the general field order f = 2j + half (positions < F of the wrapped
  // rows; wsn rows f >= F are zero), four steps per group, the next group's operands in flight behind this group's MFMAs
  f32x16 t[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) t[nb][i] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(wsn, (long)2 * JTG * 128 * 4);
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(cvec, (long)kQtConst * 4);
  float xs4[2][4], cs4[2][4];
  f32x4s ws4[2][4];
  auto ldgroup = [&](int j0, int b) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = __builtin_amdgcn_readfirstlane(j0) + u;
      xs4[b][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vhalf, j * 512, 0));
      cs4[b][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, half * 4, j * 8, 0));
      ws4[b][u] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rs, wo, j * 1024, 0));
    }
  };
  float lin = 0.f;
  ldgroup(0, 0);
#pragma unroll 1
  for (int j0 = 0; j0 < JTG; j0 += 8) {
    ldgroup(j0 + 4, 1);   // (past JTG: reads beyond the descriptors return zeros)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = 2 * (j0 + u) + half < F;
      lin = fmaf(in ? xs4[0][u] : 0.f, in ? cs4[0][u] : 0.f, lin);
      const float a = xs4[0][u];
      t[0] = mfma32(a, ws4[0][u][0], t[0]);
      t[1] = mfma32(a, ws4[0][u][1], t[1]);
      t[2] = mfma32(a, ws4[0][u][2], t[2]);
      t[3] = mfma32(a, ws4[0][u][3], t[3]);
    }
    if (j0 + 4 < JTG) {
      ldgroup(j0 + 8, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = 2 * (j0 + 4 + u) + half < F;
        lin = fmaf(in ? xs4[1][u] : 0.f, in ? cs4[1][u] : 0.f, lin);
        const float a = xs4[1][u];
        t[0] = mfma32(a, ws4[1][u][0], t[0]);
        t[1] = mfma32(a, ws4[1][u][1], t[1]);
        t[2] = mfma32(a, ws4[1][u][2], t[2]);
        t[3] = mfma32(a, ws4[1][u][3], t[3]);
      }
    }
  }
  lin = lane_halves_sum(lin);
  if (half == 0) lin_s[wave][r] = lin;
  __builtin_amdgcn_wave_barrier();

  // ---- epilogue: + b1, store x1 and R ([M][HS]; lane r owns columns 4r..4r+3), the three sum-pools
  float bv[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bv[nb] = 4 * r + nb < H ? bias1[4 * r + nb] : 0.f;
  const float cp = cvec[kQtConst + 1], cL = cvec[kQtConst];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = mfma32_row(reg, half);
    const int m = wrow0 + row;
    const float a0 = acc[0][reg] + bv[0], a1 = acc[1][reg] + bv[1], a2 = acc[2][reg] + bv[2], a3 = acc[3][reg] + bv[3];
    const float b0 = acc[4][reg], b1 = acc[5][reg], b2 = acc[6][reg], b3 = acc[7][reg];
    if (m < M) {
      *reinterpret_cast<float4*>(x1T + (long)m * HS + 4 * r) = make_float4(a0, a1, a2, a3);
      *reinterpret_cast<float4*>(RT + (long)m * HS + 4 * r) = make_float4(b0, b1, b2, b3);
    }
    float p1 = (a0 + a1) + (a2 + a3);
    float eL = a0 * b0, ep = a0 * t[0][reg];
    eL = fmaf(a1, b1, eL);
    eL = fmaf(a2, b2, eL);
    eL = fmaf(a3, b3, eL);
    ep = fmaf(a1, t[1][reg], ep);
    ep = fmaf(a2, t[2][reg], ep);
    ep = fmaf(a3, t[3][reg], ep);
    p1 = half_wave_sum_hi(p1);
    eL = half_wave_sum_hi(eL);
    ep = half_wave_sum_hi(ep);
    if (r == 31 && m < M) {
      const float vp = ep + cp, vL = (eL + lin_s[wave][row]) + cL;
      if (hf.pooled != nullptr) {   // pooled [B][L K] directly (row m = sample m >> kshift, position m & (K - 1))
        float* pr = hf.pooled + (long)(m >> hf.kshift) * hf.LK + (m & ((1 << hf.kshift) - 1));
        pr[0] = p1;
        pr[hf.op] = vp;
        pr[hf.oL] = vL;
        pv_s[wave][0][row] = p1;
        pv_s[wave][1][row] = vp;
        pv_s[wave][2][row] = vL;
      } else {
        pool1[m] = p1;
        pool_p[m] = vp;
        pool_L[m] = vL;
      }
    }
  }
  if (hf.out == nullptr) return;
  // ---- the dense head of this wave's 32 >> kshift samples: out[b] = sum_j pooled[b, j] dense_w[j] + dense_b, the products first and
  // then their sum in index order -- the arithmetic of cin_head_fwd_kernel, bit for bit
  __builtin_amdgcn_wave_barrier();
  {
    const int k = (wrow0 + r) & ((1 << hf.kshift) - 1);
    const int o0 = half == 0 ? 0 : hf.op;
    pv_s[wave][half][r] *= hf.dense_w[o0 + k];
    if (half == 0) pv_s[wave][2][r] *= hf.dense_w[hf.oL + k];
  }
  __builtin_amdgcn_wave_barrier();
  const int K = 1 << hf.kshift;
  if (lane < (32 >> hf.kshift) && wrow0 + lane * K < M) {
    float o = 0.f;
    for (int l = 0; l < 3; ++l)
      for (int k = 0; k < K; ++k) o += pv_s[wave][l][lane * K + k];
    hf.out[(wrow0 >> hf.kshift) + lane] = o + hf.dense_b[0];
  }
}

// Wave = 32 rows x 256 columns, two waves per SIMD (after the wrapped, position-major x rows took the fragment loads off the vector
// memory pipeline, 32-row waves run the 128-column kernel 4 % faster than 64-row ones: the second wave covers prologue and
// epilogue).  The generated operand and the x fragment of a step are paid once for both column halves, and all three sum-pools come
// out of the epilogue, where x1 and R of a row sit in the same lane and register:
//   pool_1[m] = sum_n x1[m,n]          pool_L[m] = sum_n x1[m,n] R[m,n] + <x[m], c> + const_L
//   pool_p[m] = sum_n x1[m,n] S[m,n] + const_p,   S = x wsum_p^T  (one more small MFMA product from the wrapped rows)
// i.e. cin_qtail_pool2_kernel's work without its pass over x1 and R.  x1 and R are still stored (the backward needs both).
// Operand layouts: W1f / WTf as cin_pack_wf_sym_body packs them ([h][2 JT][128]); wsn = wsum_p in the forward operand layout
// ([2 JTG][128], JTG = cin_jt_of(F) steps); cvec as cin_qtail_t_kernel leaves it.
template <int JT>
__global__ __launch_bounds__(256, 2) void cin_fwdq_kernel(const float* __restrict__ x2T, int XL, const float* __restrict__ W1f, const float* __restrict__ WTf,
                                                          const float* __restrict__ bias1, const float* __restrict__ wsn, int JTG,
                                                          const float* __restrict__ cvec, float* __restrict__ x1T, float* __restrict__ RT, int HS,
                                                          float* __restrict__ pool1, float* __restrict__ pool_p, float* __restrict__ pool_L, int M, int F,
                                                          int H, CinHeadFold hf) {
  constexpr int DEPTH = JT % 5 == 0 ? 5 : (JT % 4 == 0 ? 4 : (JT % 7 == 0 ? 7 : (JT % 3 == 0 ? 3 : 2)));
  static_assert(JT % DEPTH == 0, "queue depth must divide the steps per h");
  __shared__ float lin_s[4][32];
  __shared__ float pv_s[4][3][32];   // (head folded in) the wave's pooled values, then their products with dense_w
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M) return;   // (no workgroup barriers in this kernel)
  const int wrow_u = __builtin_amdgcn_readfirstlane(wrow0);
  // this wave's half of its 64-row block of the wrapped rows ([p][64 rows])
  const __amdgpu_buffer_rsrc_t rx = make_rsrc_uniform(x2T + (long)(wrow_u >> 6) * XL * 64, (long)XL * 256);
  const long wbytes = (long)F * (2 * JT) * 128 * 4;
  const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(W1f, wbytes), rwt = make_rsrc(WTf, wbytes);   // steps past the end read zeros
  const int wo = (half * 32 + r) * 16;
  const int vrow = ((wrow_u & 63) + r) * 4, vhalf = vrow + half * 256;
  auto ldw = [&](const __amdgpu_buffer_rsrc_t& rw, int s) {
    return __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rw, wo, s * 1024, 0));
  };
  auto ldfrag = [&](int h, float (&xf)[JT], float& xp) {
    const int hb = __builtin_amdgcn_readfirstlane(h) * 256;
    xp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vrow, hb, 0));
#pragma unroll
    for (int j = 0; j < JT; ++j) xf[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vhalf + 512 * j, hb, 0));
  };
  f32x16 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  f32x4s q1[DEPTH], qt[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    q1[d] = ldw(rw1, d);
    qt[d] = ldw(rwt, d);
  }
  float xa[JT], xb[JT], pa, pb;
  ldfrag(0, xa, pa);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    settle(q1[d]);
    settle(qt[d]);
  }
  settle(pa);
  float ac = pa * xa[0], an;
  auto run_h = [&](int h, float (&xc)[JT], float& pc, float (&xn_)[JT], float& pn) {
    ldfrag(min(h + 1, F - 1), xn_, pn);
    const int sb = h * JT;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const f32x4s w1 = q1[j % DEPTH], wt = qt[j % DEPTH];
      an = j + 1 < JT ? pc * xc[j + 1 < JT ? j + 1 : 0] : pn * xn_[0];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma32(ac, w1[0], acc[0]);
      acc[1] = mfma32(ac, w1[1], acc[1]);
      acc[2] = mfma32(ac, w1[2], acc[2]);
      acc[3] = mfma32(ac, w1[3], acc[3]);
      acc[4] = mfma32(ac, wt[0], acc[4]);
      acc[5] = mfma32(ac, wt[1], acc[5]);
      acc[6] = mfma32(ac, wt[2], acc[6]);
      acc[7] = mfma32(ac, wt[3], acc[7]);
      ac = an;
      __builtin_amdgcn_sched_barrier(0);
      q1[j % DEPTH] = ldw(rw1, sb + j + DEPTH);   // (after the step's MFMAs: they may land in the registers they replace)
      qt[j % DEPTH] = ldw(rwt, sb + j + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int h = 0;
#pragma unroll 1
  for (; h + 1 < F; h += 2) {
    run_h(h, xa, pa, xb, pb);
    run_h(h + 1, xb, pb, xa, pa);
  }
  if (h < F) run_h(h, xa, pa, xb, pb);

  cin_fwdq_epilogue<4>(acc, rx, vhalf, wo, r, half, wave, wrow0, bias1, wsn, JTG, cvec, x1T, RT, HS, pool1, pool_p, pool_L, M, F, H, hf, lin_s, pv_s);
}

bool cin_launch_fwdq(hipStream_t st, int JT, const float* x2T, int XL, const float* W1f, const float* WTf, const float* bias1, const float* wsn, int JTG,
                     const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf);

}  // namespace fil
