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
#include "cin_split.h"

namespace fil {

// global -> LDS staging with all of a thread's loads in flight before its first store (a plain copy loop is one exposed L2 round
// trip per element and thread: these kernels are small enough for that to be most of their time)
__device__ __forceinline__ void qt_stage(const float* __restrict__ src, float* dst, int n) {
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < n ? src[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 256 < n) dst[i0 + u * 256] = v[u];
  }
}
// rows of `cols` floats -> rows of ld floats (ld > cols: bank padding)
__device__ __forceinline__ void qt_stage_rows(const float* __restrict__ src, float* dst, int rows, int cols, int ld) {
  const int n = rows * cols;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < n ? src[i0 + u * 256] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      if (i < n) {
        const int r = i / cols;
        dst[r * ld + (i - r * cols)] = v[u];
      }
    }
  }
}

// row sums of W [C][H] (8 rows per workgroup) -> wsum[C]; wsn != nullptr: also in the forward kernels' operand layout (as
// cin_wsum_wsn_kernel: wsn[chunk][fp < JT2][128] = wsum[(chunk*128 + col)*F + fp], padding zeroed by a stride pass over `nblocks`)
__device__ __forceinline__ void qt_wsum_body(const float* __restrict__ W, float* __restrict__ wsum, int C, int H, float* __restrict__ wsn, int Hp,
                                             int F, int JT2, int chunks, int bid, int nblocks) {
  const int row = bid * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  float t = 0.f;
  if (row < C)
    for (int n = l; n < H; n += 32) t += W[(long)row * H + n];
  t = half_wave_sum(t);
  if (row < C && l == 0) {
    wsum[row] = t;
    if (wsn != nullptr) {
      const int n = row / F, fp = row - n * F;
      wsn[(((n >> 7) * JT2) + fp) * 128 + (n & 127)] = t;
    }
  }
  if (wsn != nullptr) {
    const int total = chunks * JT2 * 128;
    for (int i = bid * 256 + threadIdx.x; i < total; i += nblocks * 256) {
      const int col = i & 127, fp = (i >> 7) % JT2, chunk = (i >> 7) / JT2;
      if (fp >= F || chunk * 128 + col >= Hp) wsn[i] = 0.f;
    }
  }
}

// One launch for the forward's input-only preparation (cf. cin_fwd_prep_kernel): [0, nt) x -> xT / wrapped-row transposes; [.., +npk) the first
// layer's pair-symmetric weight pack; [.., +nwl) wsum_L; the rest (nwp workgroups): wsum_p and its MFMA operand copy.
static __global__ __launch_bounds__(256) void cin_qtail_prep_kernel(const float* __restrict__ x, float* __restrict__ xT, int F, int K, int nt,
                                                                    const float* __restrict__ W0, float* __restrict__ Wf, int H0, int JT2s, int chunks0,
                                                                    int npk, const float* __restrict__ WL, float* __restrict__ wsumL, int Hq, int HL,
                                                                    int nwl, const float* __restrict__ Wp, float* __restrict__ wsum_p,
                                                                    float* __restrict__ wsn_p, int Hpp, int JT2, int chunksp,
                                                                    float* __restrict__ x2T, int XL, int xt_in, int ks = -1, long M = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < nt) {
    if (ks >= 0) cin_transpose_block_body(x, xT, x2T, F, ks, b, M, XL, smem);   // (K = 2^ks divides 64, x as given: a workgroup per 64-row block)
    else if (xt_in) cin_wrap_rows_body(x, x2T, F, K, b, XL);
    else cin_transpose_in_body(x, xT, F, K, b, smem, x2T, XL);
  }
  else if (b < nt + npk) cin_pack_wf_sym_body(W0, Wf, F, H0, JT2s, chunks0, b - nt, npk);
  else if (b < nt + npk + nwl) qt_wsum_body(WL, wsumL, Hq * F, HL, nullptr, 0, F, 0, 0, b - nt - npk, nwl);
  else qt_wsum_body(Wp, wsum_p, Hpp * F, Hq, wsn_p, Hpp, F, JT2, chunksp, b - nt - npk - nwl, gridDim.x - nt - npk - nwl);
}

// T in both operand layouts from one launch: [0, nbf) the forward kernel's (cin_pack_wf_sym_body), [nbf, nbf + nbz) the dZ kernel's slot
// order; split-bf16 mode (nbq > 0): the rest writes the forward's planes of [W1s | Ts] (cin_qs_pack_wb_body, straight from W1 and T)
static __global__ __launch_bounds__(256) void cin_qtail_pack_kernel(const float* __restrict__ T, float* __restrict__ Wf, float* __restrict__ Wz, int F,
                                                                    int H, int JTs, int chunks, int nbf, int HS, int tiles, int nbz = -1,
                                                                    const float* __restrict__ W1 = nullptr, int H1 = 0, u32x4* __restrict__ Wb = nullptr,
                                                                    int NT = 0) {
  if (nbz < 0) nbz = gridDim.x - nbf;
  if ((int)blockIdx.x < nbf) cin_pack_wf_sym_body(T, Wf, F, H, 2 * JTs, chunks, blockIdx.x, nbf);
  else if ((int)blockIdx.x < nbf + nbz) cin_pack_wz_sym_body(T, Wz, F, H, JTs, HS, tiles, blockIdx.x - nbf, nbz);
  else cin_qs_pack_wb_body(W1, H1, T, H, Wb, NT, F, JTs, blockIdx.x - nbf - nbz, gridDim.x - nbf - nbz);
}

// LDS floats of a T workgroup (cin_qtail_t_body; fold: + the column itself) -- and of the cvec workgroup beside them
inline int cin_qtail_t_wgs(int Hpp) { return 8 * ((Hpp + 7) / 8) + 1; }   // workgroups of cin_qtail_t_body (the last one: cvec)
inline size_t cin_qtail_t_lds_floats(int F, int Hq, bool fold) {
  const size_t FT = ((size_t)F + 31) & ~(size_t)31, ldn = (((size_t)Hq + 15) & ~(size_t)15) + 1;
  return std::max(2 * FT * ldn + (fold ? (size_t)F * F : 0), (size_t)Hq * F + Hq);
}
constexpr int kQtConst = 64;   // cvec[f < F] = c[f], cvec[kQtConst] = sum_n bias_L[n], cvec[kQtConst + 1] = sum_n bias_p[n]

// Where a T workgroup puts its column in the two operand layouts itself (Wf != nullptr: no pack launch behind it; the workgroup of column h
// holds the whole column): Wf as cin_pack_wf_sym_body (JT2 = 2 JTs, chunks), Wz as cin_pack_wz_sym_body (JTs, NCOL, tiles).
struct QtPackFold {
  float* Wf;
  float* Wz;
  int JTs, chunks, NCOL, tiles;
};

// T[(f'*F + f)*Hpp + h] for block h < Hpp; block Hpp writes cvec and a zero bias vector for the R GEMM.
__device__ __forceinline__ void cin_qtail_t_body(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                 const float* __restrict__ bias_p, const float* __restrict__ bias_L, int HL,
                                                 float* __restrict__ T, float* __restrict__ cvec, float* __restrict__ zbias,
                                                 int Hpp, int F, int Hq, int bid, float* smem, QtPackFold pf = QtPackFold{nullptr, nullptr, 0, 0, 0, 0}) {
  const bool fold = pf.Wf != nullptr;
  // One workgroup per column h of T (+ the cvec workgroup, bid == 8 ceil(Hpp / 8)).  Every output of a column is a 4-byte store into a
  // line that 31 other columns share: the columns are dealt so that workgroup i of XCD i % 8 (the dispatcher's round robin) takes
  // column (i % 8) ceil(Hpp / 8) + i / 8 -- the columns of a 64-byte sector meet in ONE L2 instead of in eight.
  const int per8 = (Hpp + 7) >> 3;
  const int h = bid == 8 * per8 ? Hpp : (bid & 7) * per8 + (bid >> 3);
  if (h > Hpp || (h == Hpp && bid != 8 * per8)) return;
  if (h == Hpp) {
    if (fold) {   // columns past the layer's width in both layouts are zero (none when Hpp fills its chunks)
      const int JT2 = 2 * pf.JTs, padf = pf.chunks * 128 - Hpp, padz = pf.NCOL - Hpp;
      for (int i = threadIdx.x; i < F * JT2 * padf; i += 256) {
        const int row = i / padf, n = Hpp + (i - row * padf);
        const int hh = row / JT2, d = row - hh * JT2;
        pf.Wf[((long)((n >> 7) * F + hh) * JT2 + d) * 128 + (n & 127)] = 0.f;
      }
      for (int i = threadIdx.x; i < pf.tiles * 32 * padz; i += 256) {
        const int row = i / padz;
        pf.Wz[(long)row * pf.NCOL + Hpp + (i - row * padz)] = 0.f;
      }
    }
    float* wl = smem;            // wsum_L [Hq][F]
    float* bp = wl + Hq * F;     // bias_p [Hq]
    qt_stage(wsumL, wl, Hq * F);
    qt_stage(bias_p, bp, Hq);
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
      float t = 0.f;
      for (int n = 0; n < Hq; ++n) t = fmaf(bp[n], wl[n * F + f], t);
      cvec[f] = t;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 192) {   // wave 1: sum bias_L, wave 2: sum bias_p (64 strided chains, then the lanes in order)
      const int lane = threadIdx.x & 63, which = (threadIdx.x >> 6) - 1;
      const float* b = which == 0 ? bias_L : bias_p;
      const int nb = which == 0 ? HL : Hq;
      float t = 0.f;
      for (int n = lane; n < nb; n += 64) t += b[n];
      float tot = 0.f;
      for (int i = 0; i < 64; ++i) tot += __shfl(t, i);
      if (lane == 0) cvec[kQtConst + which] = tot;
    }
    for (int i = threadIdx.x; i < Hpp; i += 256) zbias[i] = 0.f;
    return;
  }
  // T_h[f'][f] = sum_n W_p[(h,f'),n] wsum_L[(n,f)] on the matrix pipe (v_mfma_f32_32x32x2_f32; as scalar dot products out of LDS the
  // column was 2 Hq floats read per Hq FMAs and output).  LDS, zero padded to whole tiles: wp [FT][ldn] (W_p rows of this h) |
  // wlT [FT][ldn] (wsum_L transposed: [f][n]) | tn [F F] (folded pack);  FT = F rounded up to 32, ldn = Hq rounded up to 16, + 1 (odd:
  // conflict-free down a column).  Both inputs are requested before the images are cleared.
  const int FT = (F + 31) & ~31, K16 = (Hq + 15) & ~15, ldn = K16 + 1, on = F * Hq;
  float* wp = smem;
  float* wlT = smem + FT * ldn;
  float* tn = wlT + FT * ldn;
  const float* wsrc = Wp + (long)h * F * Hq;
  const float rH = 1.f / (float)Hq, rF = 1.f / (float)F;
  constexpr int NO = 20;   // loads per thread and input in the first batch (F = 39, Hq = 128: all of them)
  float av[NO], bv[NO];
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int i = threadIdx.x + u * 256;
    av[u] = i < on ? wsrc[i] : 0.f;
    bv[u] = i < on ? wsumL[i] : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * FT * ldn; i += 256) smem[i] = 0.f;
  __syncthreads();
  auto put_a = [&](int i, float v) {   // W_p[(h,f'),n] at i = f' Hq + n
    const int fp = (int)(((float)i + 0.5f) * rH);
    wp[fp * ldn + (i - fp * Hq)] = v;
  };
  auto put_b = [&](int i, float v) {   // wsum_L[(n,f)] at i = n F + f
    const int n = (int)(((float)i + 0.5f) * rF);
    wlT[(i - n * F) * ldn + n] = v;
  };
#pragma unroll
  for (int u = 0; u < NO; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i < on) {
      put_a(i, av[u]);
      put_b(i, bv[u]);
    }
  }
  for (int i = threadIdx.x + NO * 256; i < on; i += 256) {   // (larger shapes: the rest, plainly)
    put_a(i, wsrc[i]);
    put_b(i, wsumL[i]);
  }
  __syncthreads();
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
    const int ntile = FT >> 5;   // 1 or 2 strips each way: up to four 32 x 32 tiles, one per wave
    if (wave < ntile * ntile) {
      const int mt = wave / ntile, nt = wave - mt * ntile;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ar = wp + (mt * 32 + r) * ldn + half;
      const float* br = wlT + (nt * 32 + r) * ldn + half;
      for (int k0 = 0; k0 < K16; k0 += 16) {   // eight steps' fragments read together, then their products
        float fa[8], fb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          fa[u] = ar[k0 + 2 * u];
          fb[u] = br[k0 + 2 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma32(fa[u], fb[u], acc);
      }
      const int f = nt * 32 + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int fp = mt * 32 + mfma32_row(reg, half);
        if (fp < F && f < F) {
          T[(long)(fp * F + f) * Hpp + h] = acc[reg];
          if (fold) tn[fp * F + f] = acc[reg];
        }
      }
    }
  }
  if (!fold) return;
  __syncthreads();
  // the pair weight of (hh, f = (hh + d) mod F) as the pack kernels form it
  auto sym = [&](int hh, int d) -> float {
    if (hh >= F || d > F / 2) return 0.f;
    int f = hh + d;
    if (f >= F) f -= F;
    if (d == 0) return tn[hh * F + hh];
    const float v = tn[hh * F + f] + tn[f * F + hh];
    return 2 * d == F ? 0.5f * v : v;
  };
  {
    const int JT2 = 2 * pf.JTs;
    const float rj = 1.f / (float)JT2;
    float* dst = pf.Wf + (long)(h >> 7) * F * JT2 * 128 + (h & 127);
    for (int e = threadIdx.x; e < F * JT2; e += 256) {
      const int hh = (int)(((float)e + 0.5f) * rj), d = e - hh * JT2;
      dst[(long)e * 128] = sym(hh, d);
    }
  }
  {
    const float rj = 1.f / (float)pf.JTs;
    for (int row = threadIdx.x; row < pf.tiles * 32; row += 256) {
      const int i = row & 31, t = row >> 5;
      const int rr = (i & 3) + 4 * (i >> 3), hf = (i >> 2) & 1;
      const int slot = 16 * t + rr;
      const int hh = (int)(((float)slot + 0.5f) * rj), j = slot - hh * pf.JTs;
      pf.Wz[(long)row * pf.NCOL + h] = sym(hh, 2 * j + hf);
    }
  }
}

static __global__ __launch_bounds__(256) void cin_qtail_t_kernel(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                                 const float* __restrict__ bias_p, const float* __restrict__ bias_L, int HL,
                                                                 float* __restrict__ T, float* __restrict__ cvec, float* __restrict__ zbias,
                                                                 int Hpp, int F, int Hq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cin_qtail_t_body(Wp, wsumL, bias_p, bias_L, HL, T, cvec, zbias, Hpp, F, Hq, blockIdx.x, smem);
}

// Merged quadratic-tail forward, second preparation launch: [0, nT) the T / cvec workgroups (weights only; they need wsum_L of the
// first launch), the rest: the x -> xT / wrapped-row transposes, which depend on nothing but x -- the two run side by side instead of
// one after the other (the T workgroups are the longer-running ones and come first).
static __global__ __launch_bounds__(256) void cin_qtail_t_x_kernel(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                                   const float* __restrict__ bias_p, const float* __restrict__ bias_L, int HL,
                                                                   float* __restrict__ T, float* __restrict__ cvec, float* __restrict__ zbias,
                                                                   int Hpp, int F, int Hq, int nT, const float* __restrict__ x, float* __restrict__ xT,
                                                                   int K, float* __restrict__ x2T, int XL, int xt_in, int ks = -1, long M = 0,
                                                                   QtPackFold pf = QtPackFold{nullptr, nullptr, 0, 0, 0, 0}) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x;
  if (b < nT) {
    cin_qtail_t_body(Wp, wsumL, bias_p, bias_L, HL, T, cvec, zbias, Hpp, F, Hq, b, smem, pf);
  } else if (ks >= 0) {   // (K = 2^ks divides 64, x as given: a workgroup per 64-row block)
    cin_transpose_block_body(x, xT, x2T, F, ks, b - nT, M, XL, smem);
  } else if (xt_in && M > 0) {   // (x arrives transposed: the wrapped rows by 64-row blocks)
    cin_wrap_block_body(x, x2T, F, b - nT, M, XL, smem);
  } else if (xt_in) {
    cin_wrap_rows_body(x, x2T, F, K, b - nT, XL);
  } else {
    cin_transpose_in_body(x, xT, F, K, b - nT, smem, x2T, XL);
  }
}

// Both sum-pools of the quadratic tail from ONE pass over x1 and R (MFMA form of the pooled-weights shortcut, cf. cin_last_bwd2_kernel):
//   S[m,n]   = sum_f x[m,f] wsum_p[(n,f)]   (A = the lane's x fragment, B = wsn: wsum_p in the forward kernel's operand layout)
//   pool_p[m] = sum_n x1[m,n] S[m,n] + cvec[kQtConst + 1]
//   pool_L[m] = sum_n x1[m,n] R[m,n] + sum_f x[m,f] cvec[f] + cvec[kQtConst]
// Wave = 32 rows; accumulator register `reg` of lane (r, half) is row mfma32_row(reg, half), columns 4r..4r+3.  Hp <= 128.
template <int JT>
__global__ __launch_bounds__(256, 2) void cin_qtail_pool2_kernel(const float* __restrict__ xT, const float* __restrict__ xpT, int xps,
                                                                 const float* __restrict__ wsn, const float* __restrict__ R, int HSr,
                                                                 const float* __restrict__ cvec, float* __restrict__ pool_p,
                                                                 float* __restrict__ pool_L, int M, int F, int Hp) {
  __shared__ float lin_s[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * 32;
  if (wrow0 >= M) return;
  const long mq = min(wrow0 + r, M - 1);
  float xr[JT];
  float lin = 0.f;
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int f = 2 * j + half;
    xr[j] = f < F ? xT[mq * F + f] : 0.f;
    lin = fmaf(xr[j], f < F ? cvec[f] : 0.f, lin);
  }
  lin = lane_halves_sum(lin);
  if (half == 0) lin_s[wave][r] = lin;
  f32x16 t[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) t[nb][i] = 0.f;
  const float4* wsb = reinterpret_cast<const float4*>(wsn) + (half * 32 + r);
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
  __builtin_amdgcn_wave_barrier();
  const float cp = cvec[kQtConst + 1], cL = cvec[kQtConst];
  const int n0 = 4 * r;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = mfma32_row(reg, half);
    const int m = wrow0 + row;
    const long mc = min(m, M - 1);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 < Hp) {   // (rows of a feature map are padded to whole 128-column chunks: columns >= Hp read zeros)
      a = *reinterpret_cast<const float4*>(xpT + mc * xps + n0);
      b = *reinterpret_cast<const float4*>(R + mc * HSr + n0);
    }
    float ep = a.x * t[0][reg], eL = a.x * b.x;
    ep = fmaf(a.y, t[1][reg], ep);
    ep = fmaf(a.z, t[2][reg], ep);
    ep = fmaf(a.w, t[3][reg], ep);
    eL = fmaf(a.y, b.y, eL);
    eL = fmaf(a.z, b.z, eL);
    eL = fmaf(a.w, b.w, eL);
    ep = half_wave_sum_hi(ep);
    eL = half_wave_sum_hi(eL);
    if (r == 31 && m < M) {
      pool_p[m] = ep + cp;
      pool_L[m] = (eL + lin_s[wave][row]) + cL;
    }
  }
}

// xs[m][F+1]: xs[m,f] = dP_L[m] x[m,f] (the scaled factor of the pair products in the dT GEMM), xs[m,F] = dP_p[m] (the scale of the
// F extra channels that give the shortcut's rank-one dW_p part); and, per block of 256 rows, the column sums
// dcpart[blk][f] = sum_m xs[m,f] (-> dc[f] = d pool_L / d c[f]), dcpart[blk][F] = sum_m dP_L[m], dcpart[blk][F+1] = sum_m dP_p[m].
// One row per thread.  LDS: [256][F+3]
static __global__ __launch_bounds__(256) void cin_qtail_scale_kernel(const float* __restrict__ xT, const float* __restrict__ dPL,
                                                                     const float* __restrict__ dPp, int ldp, int K, float* __restrict__ xs,
                                                                     float* __restrict__ dcpart, int M, int F, int nscale,
                                                                     const float* __restrict__ hpart, float* __restrict__ ddw, float* __restrict__ ddb,
                                                                     int LK, int nhp, int nhead, const float* __restrict__ W0, float* __restrict__ Wz,
                                                                     int H0, int JTs, int HS0, int tiles0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= nscale + nhead) {   // the first layer's weights in the dZ kernel's slot order (nothing else uses that buffer here)
    cin_pack_wz_sym_body(W0, Wz, F, H0, JTs, HS0, tiles0, blockIdx.x - nscale - nhead, gridDim.x - nscale - nhead);
    return;
  }
  if ((int)blockIdx.x >= nscale) {   // the dense head's partial sums -> ddense_w | ddense_b (fixed order)
    cin_reduce_body(hpart, ddw, (long)LK + 1, nhp, ddb, (long)LK, blockIdx.x - nscale);
    return;
  }
  const int ld = F + 3;
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  float* row = smem + threadIdx.x * ld;
  if (m < M) {
    const long b = m / K;
    const float dl = dPL[b * ldp + (m - b * K)], dp = dPp[b * ldp + (m - b * K)];
    for (int f = 0; f < F; ++f) {
      const float v = xT[m * F + f] * dl;
      xs[m * (F + 1) + f] = v;
      row[f] = v;
    }
    xs[m * (F + 1) + F] = dp;
    row[F] = dl;
    row[F + 1] = dp;
  } else {
    for (int f = 0; f < F + 2; ++f) row[f] = 0.f;
  }
  __syncthreads();
  // column sums: wave q takes rows 64q .. 64q+63 of column f = lane, the four partial sums meet in wave order
  __shared__ float cs[4][64];
  {
    const int f = threadIdx.x & 63, q = threadIdx.x >> 6;
    float t0 = 0.f, t1 = 0.f;
    if (f < F + 2) {
      for (int rr = 64 * q; rr < 64 * q + 64; rr += 2) {
        t0 += smem[rr * ld + f];
        t1 += smem[(rr + 1) * ld + f];
      }
    }
    cs[q][f] = t0 + t1;
  }
  __syncthreads();
  if (threadIdx.x < F + 2) dcpart[(long)blockIdx.x * kQtConst + threadIdx.x] = (cs[0][threadIdx.x] + cs[1][threadIdx.x]) + (cs[2][threadIdx.x] + cs[3][threadIdx.x]);
}

// Two workgroups per h < Hpp (phase = blockIdx & 1):
//   phase 0:  dW_p[(h,f'),n] = v[(h,f')] + sum_f dT[(f',f),h] wsum_L[(n,f)]       (v: the pooled-weights shortcut's rank-one part,
//   phase 1:  partL[h][(n,f)] = sum_f' W_p[(h,f'),n] dT[(f',f),h]                   given transposed, vT[f'][Hpp])
static __global__ __launch_bounds__(256) void cin_qtail_params_kernel(const float* __restrict__ Wp, const float* __restrict__ wsumL,
                                                                      const float* __restrict__ dT, const float* __restrict__ vT,
                                                                      float* __restrict__ dWp, float* __restrict__ partL, int Hpp, int F, int Hq,
                                                                      const float* __restrict__ dcpart, int ndc, float* __restrict__ dcfin) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x == 2 * Hpp) {
    // the extra workgroup: dc[f] (and the two dP sums behind it) = fixed-order sum of the block partials, for cin_qtail_fill_kernel
    float* red = smem;   // [4][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (sixteen partials in flight per lane: four at a time this one workgroup was ndc / 16 L2 round trips in a row -- the longest
    // chain of the launch)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (lane < F + 2) {
      int p = wave;
      for (; p + 60 < ndc; p += 64) {
        float u[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) u[e] = dcpart[(long)(p + 4 * e) * kQtConst + lane];
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          a0 += u[e];
          a1 += u[e + 1];
          a2 += u[e + 2];
          a3 += u[e + 3];
        }
      }
      for (; p < ndc; p += 4) a0 += dcpart[(long)p * kQtConst + lane];
    }
    red[wave * 64 + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0) dcfin[lane] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    return;
  }
  const int h = blockIdx.x >> 1, phase = blockIdx.x & 1;
  // Both products run on the matrix pipe (v_mfma_f32_32x32x2_f32, one 32-row strip of outputs per wave and visit): as scalar dot
  // products out of LDS they were 2 F floats read per F FMAs -- 12 us of LDS reads per workgroup.  LDS (zero padded to whole tiles):
  //   dt [FT][ldd]: dT[(f',f), h] as [f'][f], FT = F rounded up to 32, ldd = FT + 1 (odd: conflict-free down a column)
  //   phase 0: wl [HT][ldd]: wsum_L as [n][f], HT = Hq rounded up to 32      phase 1: wp [FK16][HT]: W_p[(h,f'),n] as [f'][n], FK16 = F rounded up to 16 (the k loops' step)
  const int FT = (F + 31) & ~31, HT = (Hq + 31) & ~31, FK16 = (F + 15) & ~15, ldd = FT + 1;
  float* dt = smem;
  float* op = dt + FT * ldd;
  const int nop = phase == 0 ? HT * ldd : FK16 * HT;
  // Both inputs are requested BEFORE the image is cleared (the clear and its barrier run under the loads): a column of dT -- one dword
  // per 512-byte row -- and the operand's F Hq contiguous floats; quotients by multiplication (indices < 2^14: exact).
  const float* osrc = phase == 0 ? wsumL : Wp + (long)h * F * Hq;
  const int ocols = phase == 0 ? F : Hq, old = phase == 0 ? ldd : HT, on = F * Hq;
  const float rF = 1.f / (float)F, rO = 1.f / (float)ocols;
  constexpr int ND = 8, NO = 24;   // loads per thread in the first batch (F = 39, Hq = 128: all of them)
  float dv[ND], ov[NO];
#pragma unroll
  for (int u = 0; u < ND; ++u) dv[u] = (int)threadIdx.x + u * 256 < F * F ? dT[(long)(threadIdx.x + u * 256) * Hpp + h] : 0.f;
#pragma unroll
  for (int u = 0; u < NO; ++u) ov[u] = (int)threadIdx.x + u * 256 < on ? osrc[threadIdx.x + u * 256] : 0.f;
  for (int i = threadIdx.x; i < FT * ldd + nop; i += 256) smem[i] = 0.f;
  __syncthreads();
  auto put_dt = [&](int i, float v) {
    const int fp = (int)(((float)i + 0.5f) * rF);
    dt[fp * ldd + (i - fp * F)] = v;
  };
  auto put_op = [&](int i, float v) {
    const int rr = (int)(((float)i + 0.5f) * rO);
    op[rr * old + (i - rr * ocols)] = v;
  };
#pragma unroll
  for (int u = 0; u < ND; ++u)
    if ((int)threadIdx.x + u * 256 < F * F) put_dt(threadIdx.x + u * 256, dv[u]);
#pragma unroll
  for (int u = 0; u < NO; ++u)
    if ((int)threadIdx.x + u * 256 < on) put_op(threadIdx.x + u * 256, ov[u]);
  for (int i = threadIdx.x + ND * 256; i < F * F; i += 256) put_dt(i, dT[(long)i * Hpp + h]);   // (larger shapes: the rest, plainly)
  for (int i = threadIdx.x + NO * 256; i < on; i += 256) put_op(i, osrc[i]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, half = lane >> 5;
  const bool two = F > 32;   // a second strip of 32 f' (phase 0) / f (phase 1)
  if (phase == 0) {
    // out[f'][n] = sum_f dt[f'][f] wl[n][f]: A = dt (row f' on the lane), B = wl (column n on the lane); a wave takes 32 columns n
    for (int nt = wave; nt < (HT >> 5); nt += 4) {
      // the shortcut's rank-one part of the lane's 16 (32) rows: requested before the products, used after them
      float v0[16], v1[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int fp = mfma32_row(reg, half);
        v0[reg] = vT[(long)min(fp, F - 1) * Hpp + h];
        v1[reg] = two ? vT[(long)min(fp + 32, F - 1) * Hpp + h] : 0.f;
      }
      f32x16 a0, a1;
#pragma unroll
      for (int i = 0; i < 16; ++i) a0[i] = a1[i] = 0.f;
      const float* ar = dt + r * ldd + half;
      const float* br = op + (nt * 32 + r) * ldd + half;
      for (int k0 = 0; k0 < FK16; k0 += 16) {   // eight steps' fragments read together, then their products (the padding is zero)
        float fa[8], fb[8], fc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          fa[u] = ar[k0 + 2 * u];
          fb[u] = br[k0 + 2 * u];
          fc[u] = two ? ar[32 * ldd + k0 + 2 * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 = mfma32(fa[u], fb[u], a0);
          if (two) a1 = mfma32(fc[u], fb[u], a1);
        }
      }
      const int n = nt * 32 + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int fp = mfma32_row(reg, half);
        if (n < Hq && fp < F) dWp[((long)h * F + fp) * Hq + n] = v0[reg] + a0[reg];
        if (two && n < Hq && fp + 32 < F) dWp[((long)h * F + fp + 32) * Hq + n] = v1[reg] + a1[reg];
      }
    }
  } else {
    // out[n][f] = sum_f' wp[f'][n] dt[f'][f]: A = wp (row n on the lane), B = dt (column f on the lane); a wave takes 32 rows n
    float* pl = partL + (long)h * Hq * F;
    for (int mt = wave; mt < (HT >> 5); mt += 4) {
      f32x16 a0, a1;
#pragma unroll
      for (int i = 0; i < 16; ++i) a0[i] = a1[i] = 0.f;
      const float* ar = op + half * HT + mt * 32 + r;
      const float* br = dt + half * ldd + r;
      for (int k0 = 0; k0 < FK16; k0 += 16) {
        float fa[8], fb[8], fc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          fa[u] = ar[(k0 + 2 * u) * HT];
          fb[u] = br[(k0 + 2 * u) * ldd];
          fc[u] = two ? br[(k0 + 2 * u) * ldd + 32] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 = mfma32(fa[u], fb[u], a0);
          if (two) a1 = mfma32(fa[u], fc[u], a1);
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = mt * 32 + mfma32_row(reg, half);
        if (n < Hq && r < F) pl[n * F + r] = a0[reg];
        if (two && n < Hq && r + 32 < F) pl[n * F + r + 32] = a1[reg];
      }
    }
  }
}

// LDS bytes of cin_qtail_params_kernel
inline size_t cin_qtail_params_lds(int F, int Hq) {
  const size_t FT = ((size_t)F + 31) & ~(size_t)31, HT = ((size_t)Hq + 31) & ~(size_t)31, FK16 = ((size_t)F + 15) & ~(size_t)15, ldd = FT + 1;
  const size_t fl = FT * ldd + (HT * ldd > FK16 * HT ? HT * ldd : FK16 * HT);
  return (fl > 256 ? fl : 256) * sizeof(float);
}

// dwsum_L[(n,f)] = sum_h partL[h][(n,f)] + bias_p[n] dc[f]  ->  dW_L[(n,f), n'] for every n' (32 rows (n,f) per workgroup of 512
// threads: half a wave per group of every 16th h, all of a lane's loads in flight together, the 16 groups summed in order);
// workgroup 0 also finishes dbias_p[n] = sum_m dP_p[m] + sum_f wsum_L[(n,f)] dc[f] and dbias_L[n'] = sum_m dP_L[m].
// dcfin[f] = dc[f], dcfin[F] = sum_m dP_L[m], dcfin[F+1] = sum_m dP_p[m] (summed by the extra workgroup of cin_qtail_params_kernel).
constexpr int kQtFillCols = 32, kQtFillThreads = 512;
static __global__ __launch_bounds__(kQtFillThreads) void cin_qtail_fill_kernel(const float* __restrict__ partL, int Hpp, const float* __restrict__ dcfin,
                                                                               const float* __restrict__ bias_p, const float* __restrict__ wsumL,
                                                                               float* __restrict__ dWL, float* __restrict__ dbias_p,
                                                                               float* __restrict__ dbias_L, int F, int Hq, int HL) {
  __shared__ float dc[kQtConst];
  __shared__ float red[16][kQtFillCols];
  __shared__ float val[kQtFillCols];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;   // 16 groups of h
  if (threadIdx.x < kQtConst) dc[threadIdx.x] = (int)threadIdx.x < F + 2 ? dcfin[threadIdx.x] : 0.f;
  const int C = Hq * F;
  const int c = blockIdx.x * kQtFillCols + col;
  float t = 0.f;
  if (c < C) {
    int h = grp;
    for (; h + 112 < Hpp; h += 128) {
      float u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = partL[(long)(h + 16 * e) * C + c];
      t += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    for (; h < Hpp; h += 16) t += partL[(long)h * C + c];
  }
  red[grp][col] = t;
  __syncthreads();
  if (grp == 0 && c < C) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) a += red[g][col];
    const int n = c / F, f = c - n * F;
    val[col] = a + bias_p[n] * dc[f];
  }
  __syncthreads();
  // the 32 rows of this workgroup x HL columns, coalesced over the columns
  for (int i = threadIdx.x; i < kQtFillCols * HL; i += kQtFillThreads) {
    const int rr = i / HL, cc = i - rr * HL;
    const int cr = blockIdx.x * kQtFillCols + rr;
    if (cr < C) dWL[(long)cr * HL + cc] = val[rr];
  }
  if (blockIdx.x == 0) {
    for (int n = threadIdx.x; n < Hq; n += kQtFillThreads) {
      float u = dc[F + 1];
      for (int f = 0; f < F; ++f) u = fmaf(wsumL[n * F + f], dc[f], u);
      dbias_p[n] = u;
    }
    for (int n = threadIdx.x; n < HL; n += kQtFillThreads) dbias_L[n] = dc[F];
  }
}

}  // namespace fil
