// A3  xDeepFM CIN forward + backward for gfx950 (fp32 MFMA, v_mfma_f32_32x32x2_f32).
//
// Replaces CIN.call of the reference (interactive_layer.py:310-327).  Per layer the reference materialises
// the outer product Z[b,k,c=h*F+f] = x^{l-1}[b,h,k] * x[b,f,k] (0.4-1.3 GB at the north-star shape),
// transposes it twice and runs a 1x1 Conv1D = GEMM [B*K, C] x [C, H].  Here Z never exists: every kernel is
// an implicit GEMM whose Z operand is regenerated in registers from the two small per-row vectors.
//
// GEMM view: rows m = b*K + k (M = B*K), reduction c = (h,f) (C = Hp*F), columns n (H).
//   fwd   cin_fwd_kernel      X^l[m,n]  = sum_c Z[m,c] W[c,n] + bias[n]          (A = Z generated, B = W via LDS)
//   bwd   cin_bwd_dw_kernel   dW[c,n]   = sum_m Z[m,c] G[m,n]                    (A = Z^T generated, B = G via LDS)
//   bwd   cin_bwd_dz_kernel   dZ[c,m]   = sum_n W[c,n] G[m,n], consumed in registers:
//                             Gx^{l-1}[m,h] = sum_f dZ[(h,f),m] x[m,f];  dX[m,f] += sum_h dZ[(h,f),m] x^{l-1}[m,h]
// The reduction order of a GEMM is free, so each kernel picks the order that makes its generated operand
// lane-local (see the per-kernel comments).  All three are MFMA-bound: fp32 MFMA issues one 32x32x2 tile per
// 64 cycles per SIMD, so LDS/VALU work per MFMA is small by construction.
//
// MFMA 32x32x2 f32 operand maps (cdna_hip_programming.md section 3): lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator register r of lane l holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
#include "common.h"

namespace fil {

constexpr int kCinThreads = 256;  // 4 waves, one per SIMD; 2 workgroups co-reside per CU (LDS <= 80 KB, VGPR <= 256)
constexpr int kCinMaxL = 8;
constexpr int kCinMaxH = 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// =================================================================================================
// Forward layer.  Workgroup tile: 128 rows (4 waves x 32 rows) x NB*32 columns.
// Reduction order: h outer; inside one h the two wave halves take f = 2j and f = 2j+1 (j < ceil(F/2));
// an odd F is padded with one zero column.  Per step a lane forms its A value with ONE multiply:
//   A[row][half] = x^{l-1}[row,h] * x[row, 2j+half]
// x^{l-1}[row,h] is one register per h (prefetched from global), x[row,:] sits in LDS (row stride odd ->
// conflict-free), the W slab of the current h ([Fpad][NB*32]) is double-buffered in LDS.
// Epilogue: bias is the accumulator's initial value; the feature map is stored as [B,H,K]; the sum-pool
// over feature maps (reference :322) is a half-wave reduction of the accumulators -> pool_part[chunk][m].
template <int NB>
__global__ __launch_bounds__(kCinThreads, 2) void cin_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ xp,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 float* __restrict__ xout, float* __restrict__ pool_part,
                                                                 int M, int F, int K, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = NB * 32;
  const int J = (F + 1) >> 1, Fpad = 2 * J, XS = Fpad + 1;
  float* x0s = smem;                 // [128][XS]
  float* Ws = smem + 128 * XS;       // [2][Fpad][NW]
  const int slab = Fpad * NW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;
  const int rowl = wave * 32 + r;     // this lane's A-operand row inside the tile
  const int m = row0 + rowl;
  const bool mvalid = m < M;
  const int b = mvalid ? m / K : 0, k = mvalid ? m - b * K : 0;

  // ---- stage x tile: x0s[row][f] = x[b,f,k] (zero for padded f / rows past M).  idx = tid + 256*u keeps
  // row = tid & 127 fixed per thread, so one (b,k) split serves all of a thread's elements.
  {
    const int srow = tid & 127;
    const int sm = row0 + srow;
    const bool sv = sm < M;
    const int sb = sv ? sm / K : 0, sk = sv ? sm - sb * K : 0;
    const float* src = x0 + ((long)sb * F) * K + sk;
    for (int f = tid >> 7; f < Fpad; f += 2) x0s[srow * XS + f] = (sv && f < F) ? src[(long)f * K] : 0.f;
  }
  // ---- W slab of one h: [Fpad][NW] elements, prefetched into registers (issue) and written to LDS (commit)
  constexpr int NSL = 64 * NW / kCinThreads;  // Fpad <= 64
  float pw[NSL];
  auto issue_slab = [&](int h) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * kCinThreads;
      const int f = idx / NW, col = idx - f * NW;
      const int n = n0 + col;
      float v = 0.f;
      if (f < F && n < H) v = W[((long)h * F + f) * H + n];
      pw[u] = v;
    }
  };
  auto commit_slab = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * kCinThreads;
      if (idx < slab) dst[idx] = pw[u];
    }
  };
  issue_slab(0);
  commit_slab(Ws);

  // Accumulators start at zero and the bias is added once in the epilogue (as Conv1D does): seeding the chain
  // with the bias would round every small product at the bias's magnitude (measured: 15x the fp32 reference error).
  f32x16 acc[NB];
  float bv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + r;
    bv[nb] = n < H ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  }
  const float* xprow = xp + ((long)b * Hp) * K + k;
  float xpv = mvalid ? xprow[0] : 0.f;
  __syncthreads();

  for (int h = 0; h < Hp; ++h) {
    const int buf = h & 1;
    const float* wsl = Ws + buf * slab;
    float xpn = 0.f;
    const bool more = h + 1 < Hp;
    if (more) {
      if (mvalid) xpn = xprow[(long)(h + 1) * K];
      issue_slab(h + 1);
    }
    const float* xrow = x0s + rowl * XS + half;
    const float* wrow = wsl + half * NW + r;
    for (int j = 0; j < J; ++j) {
      const float a = xpv * xrow[2 * j];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32(a, wrow[(2 * j) * NW + nb * 32], acc[nb]);
    }
    if (more) commit_slab(Ws + (buf ^ 1) * slab);  // the other buffer was last read in iteration h-1 (barrier below)
    xpv = xpn;
    __syncthreads();
  }

  // ---- epilogue
  const int chunk = blockIdx.y;
  const int wrow0 = row0 + wave * 32;
  const bool kvec = (K & 3) == 0;
  float bsum = 0.f;  // sum of this chunk's biases: added to the pooled GEMM part once
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bsum += bv[nb];
  bsum = half_wave_sum(bsum);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i0 = 8 * q + 4 * half;  // rows i0..i0+3 live in registers 4q..4q+3
    if (xout != nullptr) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + r;
        if (n < H) {
          if (kvec) {
            const int mm = wrow0 + i0;
            if (mm < M) {  // M = B*K is a multiple of 4 here, so the 4 rows are valid together and share b
              const int bb = mm / K, kk = mm - bb * K;
              *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
                  make_float4(acc[nb][4 * q] + bv[nb], acc[nb][4 * q + 1] + bv[nb], acc[nb][4 * q + 2] + bv[nb],
                              acc[nb][4 * q + 3] + bv[nb]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int mm = wrow0 + i0 + e;
              if (mm < M) {
                const int bb = mm / K, kk = mm - bb * K;
                xout[((long)bb * H + n) * K + kk] = acc[nb][4 * q + e] + bv[nb];
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v += acc[nb][4 * q + e];  // padded columns hold exact zeros
      v = half_wave_sum(v) + bsum;
      const int mm = wrow0 + i0 + e;
      if (r == 0 && mm < M) pool_part[(long)chunk * M + mm] = v;
    }
  }
}

// =================================================================================================
// Backward, data path: dZ^T tile = W_tile (32 channel rows) x G^T, one wave = 32 rows m (on the lanes).
// The lane (m, half) keeps its G row in registers for the whole kernel: greg[s] = G[m, half*NH + s]
// (reduction order of the n-GEMM: half 0 takes n < NH, half 1 takes n >= NH).
// Channel order: "slots".  A lane of wave half `half` owns the channels with f = 2j + half; slot index
// s = h*J + j (J = ceil(F/2)); MFMA tile t covers slots [16t, 16t+16): accumulator register r of the lane is
// exactly slot 16t + r (row map of the 32x32 tile), so the contraction of dZ with x / x^{l-1} is lane-local:
//   gxsum      += dZ * x[m,f]            (flushed when h advances:  Gprev[m,h] = gxsum(half0) + gxsum(half1) [+ dP])
//   dxs[m][f]  += dZ * x^{l-1}[m,h]      (LDS accumulator, each (m,f) is owned by exactly one lane)
// For layer 1 (x^{l-1} == x) the flushed value is added into dxs[m][h] instead of being written out.
// W tile rows are staged in MFMA-row order into LDS (double-buffered, row stride odd), one tile = 32 x H.
template <int NHMAX>
__global__ __launch_bounds__(kCinThreads, NHMAX <= 64 ? 2 : 1) void cin_bwd_dz_kernel(
    const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ x0, const float* __restrict__ xp,
    const float* __restrict__ dPprev /* [B, ldp] slice base for layer l-1, or nullptr */, int ldp,
    float* __restrict__ Gprev /* [B,Hp,K] or nullptr when layer1 */, float* __restrict__ dX, int accumulate_dx, int layer1,
    int M, int F, int K, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int J = (F + 1) >> 1, Fpad = 2 * J, XS = Fpad + 1;
  const int NH = (H + 1) >> 1;
  const int WS = H | 1;               // odd row stride of the staged W tile
  float* x0s = smem;                  // [128][XS]
  float* dxs = smem + 128 * XS;       // [128][XS]
  float* Wt = smem + 2 * 128 * XS;    // [2][32][WS]
  const int wtile = 32 * WS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * 128;
  const int rowl = wave * 32 + r;
  const int m = row0 + rowl;
  const bool mvalid = m < M;
  const int b = mvalid ? m / K : 0, k = mvalid ? m - b * K : 0;

  for (int idx = tid; idx < 128 * XS; idx += kCinThreads) dxs[idx] = 0.f;
  {
    const int srow = tid & 127;
    const int sm = row0 + srow;
    const bool sv = sm < M;
    const int sb = sv ? sm / K : 0, sk = sv ? sm - sb * K : 0;
    const float* src = x0 + ((long)sb * F) * K + sk;
    for (int f = tid >> 7; f < Fpad; f += 2) x0s[srow * XS + f] = (sv && f < F) ? src[(long)f * K] : 0.f;
  }

  // G row of this lane (B operand of every MFMA in the kernel)
  float greg[NHMAX];
  {
    const float* grow = G + ((long)b * H) * K + k;
#pragma unroll
    for (int s = 0; s < NHMAX; ++s) {
      const int n = half * NH + s;
      greg[s] = (mvalid && s < NH && n < H) ? grow[(long)n * K] : 0.f;
    }
  }

  const int Stot = Hp * J;
  const int T = (Stot + 15) >> 4;
  // W tile loader: LDS row i <-> MFMA row i <-> (slot 16t + r', half') with half' = (i>>2)&1, r' = (i&3) + 4*(i>>3).
  // A thread's elements idx = tid + 256*u map to the same (row, column) in every tile: precompute them once.
  constexpr int NT = NHMAX / 4;  // 32*H/256 elements per thread, H <= 2*NHMAX
  int pk[NT];                    // (i << 16) | n ; -1 = past the tile
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int idx = tid + u * kCinThreads;
    const int i = idx / H, n = idx - i * H;
    pk[u] = i < 32 ? ((i << 16) | n) : -1;
  }
  float pw[NT];
  int th0 = 0, tj0 = 0;  // (h, j) of slot 16*t of the tile being issued next
  auto issue_tile = [&]() {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      float v = 0.f;
      if (pk[u] >= 0) {
        const int i = pk[u] >> 16, n = pk[u] & 0xffff;
        const int rr = (i & 3) + 4 * (i >> 3), hf = (i >> 2) & 1;
        int h = th0, j = tj0 + rr;
        while (j >= J) { j -= J; ++h; }
        const int f = 2 * j + hf;
        if (h < Hp && f < F) v = W[((long)h * F + f) * H + n];
      }
      pw[u] = v;
    }
    tj0 += 16;
    while (tj0 >= J) { tj0 -= J; ++th0; }
  };
  auto commit_tile = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
      if (pk[u] >= 0) dst[(pk[u] >> 16) * WS + (pk[u] & 0xffff)] = pw[u];
  };
  issue_tile();
  commit_tile(Wt);

  // running slot -> (h, j) and the x^{l-1} values of the current / next h
  int hcur = 0, jcur = 0;
  const float* xprow = xp + ((long)b * Hp) * K + k;
  float xpv = mvalid ? xprow[0] : 0.f;
  float xpn = (mvalid && Hp > 1) ? xprow[K] : 0.f;
  float gxsum = 0.f;
  float* dxrow = dxs + rowl * XS;
  const float* xrow = x0s + rowl * XS;
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < T;
    if (more) issue_tile();
    const float* wrow = Wt + buf * wtile + r * WS + half * NH;
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NHMAX; ++s) {
      if (s < NH) d = mfma32(wrow[s], greg[s], d);
    }
    // lane-local contraction of the 16 slots of this tile
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      if (hcur < Hp) {  // uniform: slots past the end are padding
        const int f = 2 * jcur + half;
        const float dz = d[rr];
        gxsum = fmaf(dz, xrow[f], gxsum);       // x0s pad column is zero, W pad rows are zero -> dz == 0 there
        dxrow[f] = fmaf(dz, xpv, dxrow[f]);
        if (++jcur == J) {
          // flush Gx^{l-1}[m, hcur]
          const float tot = gxsum + __shfl_xor(gxsum, 32);
          if (layer1) {
            if (half == (hcur & 1)) dxrow[hcur] += tot;
          } else if (half == 0 && mvalid) {
            float v = tot;
            if (dPprev != nullptr) v += dPprev[(long)b * ldp + k];
            Gprev[((long)b * Hp + hcur) * K + k] = v;
          }
          gxsum = 0.f;
          jcur = 0;
          ++hcur;
          xpv = xpn;
          xpn = (mvalid && hcur + 1 < Hp) ? xprow[(long)(hcur + 1) * K] : 0.f;
        }
      }
    }
    if (more) commit_tile(Wt + (buf ^ 1) * wtile);  // that buffer was last read in iteration t-1 (barrier below)
    __syncthreads();
  }

  // ---- write / accumulate dX tile (row = tid & 127 is fixed per thread)
  {
    const int srow = tid & 127;
    const int sm = row0 + srow;
    if (sm < M) {
      const int sb = sm / K, sk = sm - sb * K;
      float* dst = dX + ((long)sb * F) * K + sk;
      for (int f = tid >> 7; f < F; f += 2) {
        const float v = dxs[srow * XS + f];
        float* p = dst + (long)f * K;
        *p = accumulate_dx ? *p + v : v;
      }
    }
  }
}

// =================================================================================================
// Backward, weight path: dW[c,n] = sum_m Z[m,c] G[m,n].  Workgroup = 128 channel rows (4 waves x 32) x NB*32
// columns x one split of the m range; reduction over m in LDS tiles of 64 rows.  A lane owns channel
// c = (h_i, f_i) and regenerates A[c][m] = x^{l-1}[m,h_i] * x[m,f_i] from the staged x / x^{l-1} columns;
// B[m][n] = G[m,n] from the staged G tile (row stride odd).  Tiles are prefetched into registers while the
// previous tile is being multiplied.  Split partials are summed in fixed order by cin_reduce_kernel.
constexpr int kDwMT = 64;  // m rows per LDS tile

template <int NB>
__global__ __launch_bounds__(kCinThreads, 2) void cin_bwd_dw_kernel(const float* __restrict__ G, const float* __restrict__ x0,
                                                                    const float* __restrict__ xp, float* __restrict__ part,
                                                                    int B, int F, int K, int Hp, int H, int bchunk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = NB * 32;
  constexpr int GS = NW + 1;
  const int XS = F | 1;
  const int HR = 127 / F + 2;          // distinct h values a 128-channel tile can touch
  float* Gs = smem;                    // [kDwMT][GS]
  float* x0s = Gs + kDwMT * GS;        // [kDwMT][XS]
  float* xps = x0s + kDwMT * XS;       // [kDwMT][HR]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int C = Hp * F;
  const int c0 = blockIdx.x * 128;
  const int n0 = blockIdx.z * 128;
  const int h_lo = c0 / F;
  const int c = c0 + wave * 32 + r;
  const bool cvalid = c < C;
  const int hi = cvalid ? c / F : h_lo, fi = cvalid ? c - hi * F : 0;
  const int hhi = hi - h_lo;
  const float cmask = cvalid ? 1.f : 0.f;

  const int b_lo = blockIdx.y * bchunk;
  const int b_hi = min(B, b_lo + bchunk);
  const long m_lo = (long)b_lo * K, m_hi = (long)b_hi * K;
  const int ntiles = (int)((m_hi - m_lo + kDwMT - 1) / kDwMT);

  // Staging: element idx = tid + 256*u of a [64 rows][cols] tile has row = tid & 63 for every u, so a thread
  // needs ONE (b,k) split per tile; fixed trip counts keep the prefetch registers statically indexed.
  constexpr int NG = kDwMT * NW / kCinThreads;  // G columns per thread per tile (col = (tid>>6) + 4u)
  constexpr int NXMAX = 16;                     // x columns per thread per tile (F <= 64)
  constexpr int NXP = 2;                        // x^{l-1} columns per thread per tile when HR <= 8
  float pg[NG];
  float px[NXMAX];
  float pxp[NXP];
  const int srow = tid & (kDwMT - 1), scol = tid >> 6;
  const bool xp_pref = HR <= 4 * NXP;           // uniform; tiny-F shapes stage x^{l-1} without prefetch

  auto issue = [&](int t) {
    const long mm = m_lo + (long)t * kDwMT + srow;
    const bool rv = mm < m_hi;
    const int bb = rv ? (int)(mm / K) : 0, kk = rv ? (int)(mm - (long)bb * K) : 0;
    const float* gsrc = G + ((long)bb * H) * K + kk;
    const float* xsrc = x0 + ((long)bb * F) * K + kk;
    const float* psrc = xp + ((long)bb * Hp) * K + kk;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int n = n0 + scol + 4 * u;
      pg[u] = (rv && n < H) ? gsrc[(long)n * K] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NXMAX; ++u) {
      const int f = scol + 4 * u;
      px[u] = (rv && f < F) ? xsrc[(long)f * K] : 0.f;
    }
    if (xp_pref) {
#pragma unroll
      for (int u = 0; u < NXP; ++u) {
        const int hh = scol + 4 * u;
        const int h = h_lo + hh;
        pxp[u] = (rv && hh < HR && h < Hp) ? psrc[(long)h * K] : 0.f;
      }
    }
  };
  auto commit = [&](int t) {
#pragma unroll
    for (int u = 0; u < NG; ++u) Gs[srow * GS + scol + 4 * u] = pg[u];
#pragma unroll
    for (int u = 0; u < NXMAX; ++u) {
      const int f = scol + 4 * u;
      if (f < F) x0s[srow * XS + f] = px[u];
    }
    if (xp_pref) {
#pragma unroll
      for (int u = 0; u < NXP; ++u) {
        const int hh = scol + 4 * u;
        if (hh < HR) xps[srow * HR + hh] = pxp[u];
      }
    } else {
      const long mm = m_lo + (long)t * kDwMT + srow;
      const bool rv = mm < m_hi;
      const int bb = rv ? (int)(mm / K) : 0, kk = rv ? (int)(mm - (long)bb * K) : 0;
      const float* psrc = xp + ((long)bb * Hp) * K + kk;
      for (int hh = scol; hh < HR; hh += 4) {
        const int h = h_lo + hh;
        xps[srow * HR + hh] = (rv && h < Hp) ? psrc[(long)h * K] : 0.f;
      }
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  if (ntiles > 0) issue(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // previous tile fully consumed
    commit(t);
    __syncthreads();
    if (t + 1 < ntiles) issue(t + 1);
#pragma unroll 4
    for (int s = 0; s < kDwMT / 2; ++s) {
      const int mrow = 2 * s + half;
      const float a = cmask * xps[mrow * HR + hhi] * x0s[mrow * XS + fi];
      const float* grow = Gs + mrow * GS + r;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32(a, grow[nb * 32], acc[nb]);
    }
  }

  // partial[split][c][n]
  float* pout = part + (long)blockIdx.y * C * H;
  const int crow0 = c0 + wave * 32;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cc = crow0 + mfma32_row(i, half);
      if (cc < C && n < H) pout[(long)cc * H + n] = acc[nb][i];
    }
  }
}

// out[i] = sum_{p < parts} part[p*n + i]   (fixed order)
__global__ __launch_bounds__(256) void cin_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int parts) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int p = 0; p < parts; ++p) t += part[(long)p * n + i];
    out[i] = t;
  }
}

// dbias partials: part[chunk][n] = sum_{b in chunk, k} G[b,n,k]   (thread <-> n)
__global__ __launch_bounds__(256) void cin_colsum_kernel(const float* __restrict__ G, float* __restrict__ part, int B, int H,
                                                         int K, int bchunk) {
  const int n = threadIdx.x;
  if (n >= H) return;
  const int b_lo = blockIdx.x * bchunk, b_hi = min(B, b_lo + bchunk);
  float t = 0.f;
  for (int b = b_lo; b < b_hi; ++b) {
    const float* p = G + ((long)b * H + n) * K;
    for (int k = 0; k < K; ++k) t += p[k];
  }
  part[(long)blockIdx.x * H + n] = t;
}

// G[b,n,k] = dP[b*ldp + k]  (top layer: the pooled gradient broadcast over feature maps)
__global__ __launch_bounds__(256) void cin_bcast_kernel(const float* __restrict__ dP, int ldp, float* __restrict__ G, int B,
                                                        int H, int K) {
  const long total = (long)B * H * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int b = (int)(i / ((long)H * K));
    G[i] = dP[(long)b * ldp + k];
  }
}

struct PoolArgs {
  const float* part[kCinMaxL];  // [chunks][M] per layer
  int chunks[kCinMaxL];
};

// pooled[b, l*K+k] = sum_chunk part_l[chunk][b*K+k];  out[b] = pooled[b,:] . dense_w + dense_b
__global__ __launch_bounds__(256) void cin_head_fwd_kernel(PoolArgs pa, const float* __restrict__ dense_w,
                                                           const float* __restrict__ dense_b, float* __restrict__ pooled,
                                                           float* __restrict__ out, int B, int K, int L) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const long M = (long)B * K;
  float o = 0.f;
  for (int l = 0; l < L; ++l) {
    for (int k = 0; k < K; ++k) {
      float v = 0.f;
      for (int ch = 0; ch < pa.chunks[l]; ++ch) v += pa.part[l][(long)ch * M + (long)b * K + k];
      pooled[(long)b * L * K + l * K + k] = v;
      if (out != nullptr) o = fmaf(v, dense_w[l * K + k], o);
    }
  }
  if (out != nullptr) out[b] = o + dense_b[0];
}

// output_dim == 1: dP[b,j] = g[b] * dense_w[j];  partial[blk][j] = sum_{b in blk} g[b]*pooled[b,j]  (j < LK),
// partial[blk][LK] = sum g[b].   One thread per j, blocks over chunks of b.
__global__ __launch_bounds__(256) void cin_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dense_w,
                                                           const float* __restrict__ pooled, float* __restrict__ dP,
                                                           float* __restrict__ part, int B, int LK, int bchunk) {
  const int j = threadIdx.x;
  if (j > LK) return;
  const int b_lo = blockIdx.x * bchunk, b_hi = min(B, b_lo + bchunk);
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

// ------------------------------------------------------------------------------------------------- host side
struct CinShape {
  int B, F, K, L;
  int H[kCinMaxL];
  int Hp(int l) const { return l == 0 ? F : H[l - 1]; }
  long M() const { return (long)B * K; }
  int Hmax() const {
    int h = 0;
    for (int l = 0; l < L; ++l) h = std::max(h, H[l]);
    return h;
  }
  long Cmax() const {
    long c = 0;
    for (int l = 0; l < L; ++l) c = std::max(c, (long)Hp(l) * F);
    return c;
  }
};

static int check_shape(const char* fn, int B, int F, int K, int L, const int* H, CinShape& s) {
  if (B < 0 || F < 1 || K < 1 || L < 1 || H == nullptr) return fail(FIL_ERR_ARG, "%s: bad shape B=%d F=%d K=%d L=%d", fn, B, F, K, L);
  if (L > kCinMaxL) return fail(FIL_ERR_UNSUPPORTED, "%s: L=%d > %d", fn, L, kCinMaxL);
  if (F > 64) return fail(FIL_ERR_UNSUPPORTED, "%s: F=%d > 64 fields", fn, F);
  if ((long)B * K > (1L << 30)) return fail(FIL_ERR_UNSUPPORTED, "%s: B*K too large", fn);
  s.B = B; s.F = F; s.K = K; s.L = L;
  for (int l = 0; l < L; ++l) {
    if (H[l] < 1) return fail(FIL_ERR_ARG, "%s: H[%d]=%d", fn, l, H[l]);
    if (H[l] > kCinMaxH) return fail(FIL_ERR_UNSUPPORTED, "%s: H[%d]=%d > %d feature maps", fn, l, H[l], kCinMaxH);
    s.H[l] = H[l];
  }
  return FIL_OK;
}

static int chunks_of(int H) { return cdiv(H, 128); }

static const char* kFwdNames[kCinMaxL] = {"cin_fwd_l1", "cin_fwd_l2", "cin_fwd_l3", "cin_fwd_l4", "cin_fwd_l5", "cin_fwd_l6", "cin_fwd_l7", "cin_fwd_l8"};
static const char* kDwNames[kCinMaxL] = {"cin_bwd_dw_l1", "cin_bwd_dw_l2", "cin_bwd_dw_l3", "cin_bwd_dw_l4", "cin_bwd_dw_l5", "cin_bwd_dw_l6", "cin_bwd_dw_l7", "cin_bwd_dw_l8"};
static const char* kDzNames[kCinMaxL] = {"cin_bwd_dz_l1", "cin_bwd_dz_l2", "cin_bwd_dz_l3", "cin_bwd_dz_l4", "cin_bwd_dz_l5", "cin_bwd_dz_l6", "cin_bwd_dz_l7", "cin_bwd_dz_l8"};
// algorithmic flops of one layer GEMM: 2 * M * C * H
static double gemm_flops(long M, int Hp, int F, int H) { return 2.0 * (double)M * Hp * F * H; }

// number of m-range splits for the dW kernel: fill ~2 workgroups per CU
static int dw_splits(const CinShape& s, int l) {
  const long C = (long)s.Hp(l) * s.F;
  const int ctiles = (int)((C + 127) / 128) * chunks_of(s.H[l]);
  int sp = std::max(1, (512 + ctiles / 2) / ctiles);
  sp = std::min(sp, std::max(1, s.B));
  sp = std::min(sp, 64);
  return sp;
}

static size_t fwd_ws_bytes(const CinShape& s) {
  size_t t = 0;
  for (int l = 0; l < s.L; ++l) t += align_up((size_t)chunks_of(s.H[l]) * s.M() * sizeof(float), 256);
  return t;
}

constexpr int kHeadChunk = 64;   // samples per block in the head / colsum partial reductions

static size_t bwd_ws_bytes(const CinShape& s) {
  size_t t = 0;
  const size_t LK = (size_t)s.L * s.K;
  t += align_up((size_t)s.B * LK * sizeof(float), 256);                                  // dP
  t += 2 * align_up((size_t)s.B * s.Hmax() * s.K * sizeof(float), 256);                  // G ping-pong
  size_t pmax = 0;
  for (int l = 0; l < s.L; ++l) pmax = std::max(pmax, (size_t)dw_splits(s, l) * s.Hp(l) * s.F * s.H[l]);
  t += align_up(pmax * sizeof(float), 256);                                              // dW partials
  const size_t nblk = (size_t)cdiv(std::max(1, s.B), kHeadChunk);
  t += align_up(nblk * std::max((size_t)s.Hmax(), LK + 1) * sizeof(float), 256);          // colsum / head partials
  return t;
}

template <typename KernelT>
static void allow_lds(KernelT kernel, size_t sh) {
  if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
}

static int launch_fwd_layer(hipStream_t st, const float* x0, const float* xp, const float* W, const float* bias, float* xout,
                            float* pool_part, int M, int F, int K, int Hp, int H) {
  const int J = (F + 1) / 2, Fpad = 2 * J, XS = Fpad + 1;
  const int chunks = chunks_of(H);
  const dim3 grid(cdiv(M, 128), chunks);
  // every chunk uses the widest tile it needs; the last chunk may be narrower
  const int nb_full = std::min(4, cdiv(H, 32));
  // use one NB for the whole launch (the widest needed): columns past H are zero-padded
  const int NB = chunks > 1 ? 4 : nb_full;
  const size_t sh = ((size_t)128 * XS + 2 * (size_t)Fpad * NB * 32) * sizeof(float);
#define FIL_FWD(NBV)                                                                                              \
  case NBV:                                                                                                       \
    allow_lds(cin_fwd_kernel<NBV>, sh);                                                                           \
    hipLaunchKernelGGL((cin_fwd_kernel<NBV>), grid, dim3(kCinThreads), sh, st, x0, xp, W, bias, xout, pool_part, M, F, K, Hp, H); \
    break;
  switch (NB) { FIL_FWD(1) FIL_FWD(2) FIL_FWD(3) FIL_FWD(4) }
#undef FIL_FWD
  return 0;
}

static int launch_dw(hipStream_t st, const float* G, const float* x0, const float* xp, float* part, int B, int F, int K, int Hp,
                     int H, int splits) {
  const int C = Hp * F;
  const int chunks = chunks_of(H);
  const int NB = chunks > 1 ? 4 : std::min(4, cdiv(H, 32));
  const int bchunk = cdiv(B, splits);
  const dim3 grid(cdiv(C, 128), cdiv(B, bchunk), chunks);
  const int XS = F | 1, HR = 127 / F + 2;
  const size_t sh = ((size_t)kDwMT * (NB * 32 + 1) + (size_t)kDwMT * XS + (size_t)kDwMT * HR) * sizeof(float);
#define FIL_DW(NBV)                                                                                                  \
  case NBV:                                                                                                          \
    allow_lds(cin_bwd_dw_kernel<NBV>, sh);                                                                           \
    hipLaunchKernelGGL((cin_bwd_dw_kernel<NBV>), grid, dim3(kCinThreads), sh, st, G, x0, xp, part, B, F, K, Hp, H, bchunk); \
    break;
  switch (NB) { FIL_DW(1) FIL_DW(2) FIL_DW(3) FIL_DW(4) }
#undef FIL_DW
  return (int)grid.y;  // actual number of partials written
}

static void launch_dz(hipStream_t st, const float* G, const float* W, const float* x0, const float* xp, const float* dPprev,
                      int ldp, float* Gprev, float* dX, int accumulate, int layer1, int M, int F, int K, int Hp, int H) {
  const int J = (F + 1) / 2, XS = 2 * J + 1;
  const int WS = H | 1;
  const size_t sh = ((size_t)2 * 128 * XS + 2 * (size_t)32 * WS) * sizeof(float);
  const dim3 grid(cdiv(M, 128));
  if (H <= 128) {
    allow_lds(cin_bwd_dz_kernel<64>, sh);
    hipLaunchKernelGGL((cin_bwd_dz_kernel<64>), grid, dim3(kCinThreads), sh, st, G, W, x0, xp, dPprev, ldp, Gprev, dX,
                       accumulate, layer1, M, F, K, Hp, H);
  } else {
    allow_lds(cin_bwd_dz_kernel<128>, sh);
    hipLaunchKernelGGL((cin_bwd_dz_kernel<128>), grid, dim3(kCinThreads), sh, st, G, W, x0, xp, dPprev, ldp, Gprev, dX,
                       accumulate, layer1, M, F, K, Hp, H);
  }
}

}  // namespace fil

using namespace fil;

extern "C" size_t fil_cin_saved_bytes(int B, int F, int K, int L, const int* H) {
  if (B <= 0 || K <= 0 || L <= 1 || H == nullptr) return 0;
  size_t t = 0;
  for (int l = 0; l + 1 < L; ++l) t += align_up((size_t)B * H[l] * K * sizeof(float), 256);
  return t;
}

extern "C" size_t fil_cin_fwd_workspace_bytes(int B, int F, int K, int L, const int* H) {
  CinShape s;
  if (check_shape("fil_cin_fwd_workspace_bytes", B, F, K, L, H, s) != FIL_OK) return 0;
  return fwd_ws_bytes(s);
}

extern "C" size_t fil_cin_bwd_workspace_bytes(int B, int F, int K, int L, const int* H) {
  CinShape s;
  if (check_shape("fil_cin_bwd_workspace_bytes", B, F, K, L, H, s) != FIL_OK) return 0;
  return bwd_ws_bytes(s);
}

extern "C" int fil_cin_fwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                           const float* dense_b, float* out, float* pooled, float* saved, int B, int F, int K, int L,
                           const int* H, int output_dim, int mode, void* workspace, size_t workspace_bytes, void* stream) {
  CinShape s;
  int rc = check_shape("fil_cin_fwd", B, F, K, L, H, s);
  if (rc != FIL_OK) return rc;
  if (mode != 0) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_fwd: mode %d (only 0 = fp32 MFMA)", mode);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(x && W && bias && pooled);
  FIL_CHECK_ARG(output_dim != 1 || (dense_w && dense_b && out));
  FIL_CHECK_ARG(L == 1 || saved != nullptr);
  if (workspace == nullptr || workspace_bytes < fwd_ws_bytes(s))
    return fail(FIL_ERR_WORKSPACE, "fil_cin_fwd: workspace %zu < %zu bytes", workspace_bytes, fwd_ws_bytes(s));
  hipStream_t st = (hipStream_t)stream;
  const int M = (int)s.M();
  Carver ws(workspace);
  PoolArgs pa;
  const float* xp = x;
  char* sv = reinterpret_cast<char*>(saved);
  for (int l = 0; l < L; ++l) {
    FIL_CHECK_ARG(W[l] && bias[l]);
    float* part = ws.take<float>((size_t)chunks_of(H[l]) * M);
    pa.part[l] = part;
    pa.chunks[l] = chunks_of(H[l]);
    float* xout = nullptr;
    if (l + 1 < L) {
      xout = reinterpret_cast<float*>(sv);
      sv += align_up((size_t)B * H[l] * K * sizeof(float), 256);
    }
    {
      ProfScope ps(kFwdNames[l], st, gemm_flops(M, s.Hp(l), F, H[l]));
      launch_fwd_layer(st, x, xp, W[l], bias[l], xout, part, M, F, K, s.Hp(l), H[l]);
    }
    FIL_CHECK_LAUNCH();
    xp = xout;
  }
  {
    ProfScope ps("cin_head_fwd", st);
    hipLaunchKernelGGL(cin_head_fwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, pa, dense_w, dense_b, pooled,
                       output_dim == 1 ? out : nullptr, B, K, L);
  }
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_cin_bwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                           const float* pooled, const float* saved, const float* g, float* dx, float* const* dW,
                           float* const* dbias, float* ddense_w, float* ddense_b, int B, int F, int K, int L, const int* H,
                           int output_dim, int mode, void* workspace, size_t workspace_bytes, void* stream) {
  (void)bias;
  CinShape s;
  int rc = check_shape("fil_cin_bwd", B, F, K, L, H, s);
  if (rc != FIL_OK) return rc;
  if (mode != 0) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_bwd: mode %d (only 0 = fp32 MFMA)", mode);
  FIL_CHECK_ARG(W && dW && dbias);
  hipStream_t st = (hipStream_t)stream;
  const size_t LK = (size_t)L * K;
  if (B == 0) {  // empty batch: parameter gradients are zero
    for (int l = 0; l < L; ++l) {
      (void)hipMemsetAsync(dW[l], 0, (size_t)s.Hp(l) * F * H[l] * sizeof(float), st);
      (void)hipMemsetAsync(dbias[l], 0, (size_t)H[l] * sizeof(float), st);
    }
    if (output_dim == 1) {
      (void)hipMemsetAsync(ddense_w, 0, LK * sizeof(float), st);
      (void)hipMemsetAsync(ddense_b, 0, sizeof(float), st);
    }
    return FIL_OK;
  }
  FIL_CHECK_ARG(x && g && dx);
  FIL_CHECK_ARG(output_dim != 1 || (dense_w && pooled && ddense_w && ddense_b));
  FIL_CHECK_ARG(L == 1 || saved != nullptr);
  if (workspace == nullptr || workspace_bytes < bwd_ws_bytes(s))
    return fail(FIL_ERR_WORKSPACE, "fil_cin_bwd: workspace %zu < %zu bytes", workspace_bytes, bwd_ws_bytes(s));
  const int M = (int)s.M();
  Carver ws(workspace);
  float* dP = ws.take<float>((size_t)B * LK);
  float* Gbuf[2];
  Gbuf[0] = ws.take<float>((size_t)B * s.Hmax() * K);
  Gbuf[1] = ws.take<float>((size_t)B * s.Hmax() * K);
  size_t pmax = 0;
  for (int l = 0; l < L; ++l) pmax = std::max(pmax, (size_t)dw_splits(s, l) * s.Hp(l) * F * H[l]);
  float* part = ws.take<float>(pmax);
  const int nblk = cdiv(B, kHeadChunk);
  float* small = ws.take<float>((size_t)nblk * std::max((size_t)s.Hmax(), LK + 1));

  // ---- head backward: dP, ddense_w, ddense_b
  const float* dPsrc = g;  // output_dim != 1: g is already dL/dpooled
  if (output_dim == 1) {
    if (LK + 1 > 256) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_bwd: L*K=%zu > 255", LK);
    hipLaunchKernelGGL(cin_head_bwd_kernel, dim3(nblk), dim3(256), 0, st, g, dense_w, pooled, dP, small, B, (int)LK, kHeadChunk);
    FIL_CHECK_LAUNCH();
    dPsrc = dP;
  }
  // saved map pointers
  const float* maps[kCinMaxL];
  {
    const char* sv = reinterpret_cast<const char*>(saved);
    for (int l = 0; l + 1 < L; ++l) {
      maps[l] = reinterpret_cast<const float*>(sv);
      sv += align_up((size_t)B * H[l] * K * sizeof(float), 256);
    }
  }
  if (output_dim == 1) {
    // small is [nblk][LK+1]; sum over blocks into a temp then split
    float* tmp = Gbuf[0];  // not yet in use
    hipLaunchKernelGGL(cin_reduce_kernel, dim3(1), dim3(256), 0, st, small, tmp, (long)(LK + 1), nblk);
    FIL_CHECK_LAUNCH();
    (void)hipMemcpyAsync(ddense_w, tmp, LK * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(ddense_b, tmp + LK, sizeof(float), hipMemcpyDeviceToDevice, st);
  }

  // ---- top layer gradient: broadcast of its pooled gradient
  int cur = 0;
  {
    const long total = (long)B * H[L - 1] * K;
    const int grid = (int)std::min<long>((total + 255) / 256, 4096);
    ProfScope ps("cin_bcast_g", st, (double)total * sizeof(float));
    hipLaunchKernelGGL(cin_bcast_kernel, dim3(grid), dim3(256), 0, st, dPsrc + (size_t)(L - 1) * K, (int)LK, Gbuf[cur], B, H[L - 1], K);
    FIL_CHECK_LAUNCH();
  }
  for (int l = L - 1; l >= 0; --l) {
    FIL_CHECK_ARG(W[l] && dW[l] && dbias[l]);
    const int Hp = s.Hp(l), Hl = H[l];
    const float* xp = l == 0 ? x : maps[l - 1];
    const float* G = Gbuf[cur];
    // dbias
    {
      ProfScope ps("cin_dbias", st, (double)B * Hl * K * sizeof(float));
      hipLaunchKernelGGL(cin_colsum_kernel, dim3(nblk), dim3(256), 0, st, G, small, B, Hl, K, kHeadChunk);
      hipLaunchKernelGGL(cin_reduce_kernel, dim3(1), dim3(256), 0, st, small, dbias[l], (long)Hl, nblk);
    }
    FIL_CHECK_LAUNCH();
    // dW
    int parts;
    {
      ProfScope ps(kDwNames[l], st, gemm_flops(M, Hp, F, Hl));
      parts = launch_dw(st, G, x, xp, part, B, F, K, Hp, Hl, dw_splits(s, l));
    }
    FIL_CHECK_LAUNCH();
    const long nW = (long)Hp * F * Hl;
    {
      ProfScope ps("cin_reduce_dw", st, (double)(parts + 1) * nW * sizeof(float));
      hipLaunchKernelGGL(cin_reduce_kernel, dim3((int)std::min<long>((nW + 255) / 256, 2048)), dim3(256), 0, st, part, dW[l], nW, parts);
    }
    FIL_CHECK_LAUNCH();
    // dZ -> G^{l-1}, dX
    const float* dPprev = l > 0 ? dPsrc + (size_t)(l - 1) * K : nullptr;
    {
      ProfScope ps(kDzNames[l], st, gemm_flops(M, Hp, F, Hl));
      launch_dz(st, G, W[l], x, xp, dPprev, (int)LK, l > 0 ? Gbuf[cur ^ 1] : nullptr, dx, /*accumulate=*/l != L - 1,
                /*layer1=*/l == 0, M, F, K, Hp, Hl);
    }
    FIL_CHECK_LAUNCH();
    cur ^= 1;
  }
  return FIL_OK;
}
