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

// Makes the compiler wait HERE for any pending load of v (an empty asm that "uses" the register), so that no
// s_waitcnt vmcnt(0) is placed later inside an MFMA loop, where it would also drain the prefetch just issued.
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }

// =================================================================================================
// Forward layer.  Workgroup tile: 128 rows (4 waves x 32 rows) x NB*32 columns.
// Reduction order: h outer; inside one h the two wave halves take f = 2j and f = 2j+1 (j < ceil(F/2));
// an odd F is padded with one zero column.  Per step a lane forms its A value with ONE multiply:
//   A[row][half] = x^{l-1}[row,h] * x[row, 2j+half]
// x^{l-1}[row,h] is one register per h (prefetched from global), x[row,:] sits in LDS (row stride odd ->
// conflict-free), the W slab of the current h ([Fpad][NB*32]) is double-buffered in LDS: slab h+1 is fetched
// into registers at the top of iteration h and written to the other buffer after the MFMA loop.
// LDS operands of step j+1 are read before the MFMAs of step j are issued (software prefetch).
// Epilogue: + bias; the feature map is stored as [B,H,K]; the sum-pool over feature maps (reference :322) is a
// half-wave reduction of the accumulators -> pool_part[chunk][m].  xout == nullptr: pooled output only.
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
  float xpn = 0.f;
  __syncthreads();

  const float* xrow = x0s + rowl * XS + half;
  for (int h = 0; h < Hp; ++h) {
    const int buf = h & 1;
    const bool more = h + 1 < Hp;
    settle(xpv);  // loads from the previous iteration are complete here; nothing waits on VMEM inside the MFMA loop
    if (more) {
      xpn = mvalid ? xprow[(long)(h + 1) * K] : 0.f;
      issue_slab(h + 1);
    }
    const float* wrow = Ws + buf * slab + half * NW + r;
    // software-prefetched operand stream
    float xa = xrow[0];
    float wb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wb[nb] = wrow[nb * 32];
#pragma unroll 2
    for (int j = 0; j < J; ++j) {
      const float a = xpv * xa;
      float wc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wc[nb] = wb[nb];
      const int jn = min(j + 1, J - 1);
      xa = xrow[2 * jn];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wb[nb] = wrow[(2 * jn) * NW + nb * 32];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32(a, wc[nb], acc[nb]);
      // keep the next step's LDS reads ahead of this step's MFMAs (the scheduler otherwise sinks them to their use)
      __builtin_amdgcn_sched_group_barrier(0x100, 1 + (NB + 1) / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
    }
    if (more) {
      commit_slab(Ws + (buf ^ 1) * slab);  // the other buffer was last read in iteration h-1 (barrier below)
      xpv = xpn;
    }
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
// The lane (m, half) keeps its G row in registers for the whole kernel: greg[s] = G[m, half*NHMAX + s]
// (reduction order of the n-GEMM: half 0 takes n < NHMAX, half 1 takes n >= NHMAX; columns >= H are zero).
// Channel order: "slots".  A lane of wave half `half` owns the channels with f = 2j + half; slot index
// s = h*J + j (J = ceil(F/2)); MFMA tile t covers slots [16t, 16t+16): accumulator register r of the lane is
// exactly slot 16t + r (row map of the 32x32 tile), so the contraction of dZ with x / x^{l-1} is lane-local:
//   gxsum      += dZ * x[m,f]            (flushed when h advances:  Gprev[m,h] = gxsum(half0) + gxsum(half1) [+ dP])
//   dxs[m][f]  += dZ * x^{l-1}[m,h]      (LDS accumulator, each (m,f) is owned by exactly one lane)
// For layer 1 (x^{l-1} == x) the flushed value is added into dxs[m][h] instead of being written out.
// W tile rows are staged in MFMA-row order into LDS (double-buffered; row stride 2*NHMAX+4 floats: 16-byte
// aligned and conflict-free for ds_read_b128 with one row per lane), one tile = 32 x 2*NHMAX.  The MFMA chain of
// a tile is straight-line code (no per-step branch): 4 steps per 16-byte LDS read.
template <int NHMAX>
__global__ __launch_bounds__(kCinThreads, NHMAX <= 64 ? 2 : 1) void cin_bwd_dz_kernel(
    const float* __restrict__ G, const float* __restrict__ W, const float* __restrict__ x0, const float* __restrict__ xp,
    const float* __restrict__ dPprev /* [B, ldp] slice base for layer l-1, or nullptr */, int ldp,
    float* __restrict__ Gprev /* [B,Hp,K] or nullptr when layer1 */, float* __restrict__ dX, int accumulate_dx, int layer1,
    int M, int F, int K, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NCOL = 2 * NHMAX;
  constexpr int WS = NCOL + 4;
  constexpr int NT = 32 * NCOL / kCinThreads;  // W tile elements per thread
  const int J = (F + 1) >> 1, Fpad = 2 * J, XS = Fpad + 1;
  const int xs_words = (2 * 128 * XS + 3) & ~3;  // keep the W tiles 16-byte aligned
  float* x0s = smem;                  // [128][XS]
  float* dxs = smem + 128 * XS;       // [128][XS]
  float* Wt = smem + xs_words;        // [2][32][WS]
  constexpr int wtile = 32 * WS;

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
      const int n = half * NHMAX + s;
      greg[s] = (mvalid && n < H) ? grow[(long)n * K] : 0.f;
    }
  }

  const int Stot = Hp * J;
  const int T = (Stot + 15) >> 4;
  // W tile loader: LDS row i <-> MFMA row i <-> (slot 16t + r', half') with half' = (i>>2)&1, r' = (i&3) + 4*(i>>3)
  float pw[NT];
  int th0 = 0, tj0 = 0;  // (h, j) of slot 16*t of the tile being issued next
  const float* xprow = xp + ((long)b * Hp) * K + k;
  float xcur[16], xnext[16];  // x^{l-1}[m, h(slot)] for the 16 slots of the current / next tile
  auto issue_tile = [&]() {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int idx = tid + u * kCinThreads;
      const int i = idx / NCOL, n = idx - i * NCOL;
      const int rr = (i & 3) + 4 * (i >> 3), hf = (i >> 2) & 1;
      int h = th0, j = tj0 + rr;
      while (j >= J) { j -= J; ++h; }
      const int f = 2 * j + hf;
      pw[u] = (h < Hp && f < F && n < H) ? W[((long)h * F + f) * H + n] : 0.f;
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      int h = th0, j = tj0 + rr;
      while (j >= J) { j -= J; ++h; }
      xnext[rr] = (mvalid && h < Hp) ? xprow[(long)h * K] : 0.f;
    }
    tj0 += 16;
    while (tj0 >= J) { tj0 -= J; ++th0; }
  };
  auto commit_tile = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int idx = tid + u * kCinThreads;
      const int i = idx / NCOL, n = idx - i * NCOL;
      dst[i * WS + n] = pw[u];
    }
  };
  issue_tile();
  commit_tile(Wt);

  int hcur = 0, jcur = 0;  // running slot -> (h, j) of the contraction
  float gxsum = 0.f;
  float* dxrow = dxs + rowl * XS;
  const float* xrow = x0s + rowl * XS;
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < T;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      xcur[rr] = xnext[rr];
      settle(xcur[rr]);  // the loads issued one iteration ago are waited for here, not inside the MFMA chain
    }
    if (more) issue_tile();
    const float* wrow = Wt + buf * wtile + r * WS + half * NHMAX;
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < NHMAX / 4; ++s4) {
      const float4 w4 = *reinterpret_cast<const float4*>(wrow + 4 * s4);
      d = mfma32(w4.x, greg[4 * s4 + 0], d);
      d = mfma32(w4.y, greg[4 * s4 + 1], d);
      d = mfma32(w4.z, greg[4 * s4 + 2], d);
      d = mfma32(w4.w, greg[4 * s4 + 3], d);
    }
    // lane-local contraction of the 16 slots of this tile
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      if (hcur < Hp) {  // uniform: slots past the end are padding
        const int f = 2 * jcur + half;
        const float dz = d[rr];
        gxsum = fmaf(dz, xrow[f], gxsum);       // x0s pad column is zero, W pad rows are zero -> dz == 0 there
        dxrow[f] = fmaf(dz, xcur[rr], dxrow[f]);
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
// previous tile is being multiplied; LDS operands of step s+1 are read before the MFMAs of step s.
// Split partials are summed in fixed order by cin_reduce_kernel.
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
      px[u] = (rv && f < F) ? (x0 != nullptr ? xsrc[(long)f * K] : 1.f) : 0.f;
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

  const float* xpcol = xps + half * HR + hhi;   // + 2s*HR
  const float* x0col = x0s + half * XS + fi;    // + 2s*XS
  const float* gcol = Gs + half * GS + r;       // + 2s*GS + nb*32
  if (ntiles > 0) issue(0);
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();  // previous tile fully consumed
    commit(t);
    __syncthreads();
    if (t + 1 < ntiles) issue(t + 1);
    float xv = xpcol[0], zv = x0col[0];
    float gb[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) gb[nb] = gcol[nb * 32];
#pragma unroll 2
    for (int s = 0; s < kDwMT / 2; ++s) {
      const float a = cmask * xv * zv;
      float gc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) gc[nb] = gb[nb];
      const int sn = min(s + 1, kDwMT / 2 - 1);
      xv = xpcol[2 * sn * HR];
      zv = x0col[2 * sn * XS];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) gb[nb] = gcol[2 * sn * GS + nb * 32];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma32(a, gc[nb], acc[nb]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + (NB + 1) / 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
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

// out[i] = sum_{p < parts} part[p*n + i]   (fixed order).  One workgroup per 64 outputs; the 4 waves take every 4th
// partial (coalesced over i), then the 4 wave sums are added in wave order -> many loads in flight, fixed order.
__global__ __launch_bounds__(256) void cin_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int parts) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < n) {
    int p = wave;
    for (; p + 12 < parts; p += 16) {
      t0 += part[(long)p * n + i];
      t1 += part[(long)(p + 4) * n + i];
      t2 += part[(long)(p + 8) * n + i];
      t3 += part[(long)(p + 12) * n + i];
    }
    for (; p < parts; p += 4) t0 += part[(long)p * n + i];
  }
  red[wave][lane] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (wave == 0 && i < n) out[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// dbias partials: part[chunk][n] = sum_{b in chunk, k} G[b,n,k].  A sample is H*K contiguous floats: thread t sums
// positions t, t+256, ... over the samples of its chunk (coalesced), the per-position sums go to LDS and thread n
// adds its K positions in order -> fixed summation order, many workgroups.
__global__ __launch_bounds__(256) void cin_colsum_kernel(const float* __restrict__ G, float* __restrict__ part, int B, int H,
                                                         int K, int bchunk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [H*K]
  const int b_lo = blockIdx.x * bchunk, b_hi = min(B, b_lo + bchunk);
  const int HK = H * K;
  for (int idx = threadIdx.x; idx < HK; idx += 256) {
    float t = 0.f;
    for (int b = b_lo; b < b_hi; ++b) t += G[(long)b * HK + idx];
    smem[idx] = t;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < H; n += 256) {
    float t = 0.f;
    for (int k = 0; k < K; ++k) t += smem[n * K + k];
    part[(long)blockIdx.x * H + n] = t;
  }
}

// part[blk] = sum over a chunk of samples of dP[b*ldp + k], k < K  (dbias of the last layer: same value for every n)
__global__ __launch_bounds__(256) void cin_slice_sum_kernel(const float* __restrict__ dP, int ldp, float* __restrict__ part, int B,
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

// dbias[n] = sum_p part[p] for every n < H
__global__ __launch_bounds__(256) void cin_fill_sum_kernel(const float* __restrict__ part, int parts, float* __restrict__ dbias, int H) {
  float t = 0.f;
  for (int p = 0; p < parts; ++p) t += part[p];
  for (int n = threadIdx.x; n < H; n += 256) dbias[n] = t;
}

// y[b,f,k] = x[b,f,k] * dP[b*ldp + k]
__global__ __launch_bounds__(256) void cin_scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ dP, int ldp,
                                                             float* __restrict__ y, int B, int F, int K) {
  const long total = (long)B * F * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int b = (int)(i / ((long)F * K));
    y[i] = x[i] * dP[(long)b * ldp + k];
  }
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

__global__ __launch_bounds__(256) void cin_wsum_kernel(const float* __restrict__ W, float* __restrict__ wsum, int C, int H) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  float t = 0.f;
  if (row < C)
    for (int n = l; n < H; n += 32) t += W[(long)row * H + n];
  t = half_wave_sum(t);
  if (row < C && l == 0) wsum[row] = t;
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

__global__ __launch_bounds__(256) void cin_last_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ xp,
                                                           const float* __restrict__ wsum, const float* __restrict__ bias,
                                                           float* __restrict__ pool, int M, int F, int K, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int FP4 = (F + 3) & ~3;
  stage_wsum(wsum, smem, Hp, F, FP4);
  float bsum = 0.f;
  for (int n = 0; n < H; ++n) bsum += bias[n];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * kLastRows + wave * 16 + (lane & 15), hq = lane >> 4;
  const bool valid = m < M;
  const int b = valid ? m / K : 0, k = valid ? m - b * K : 0;
  float xr[kLastFMax];
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) xr[f] = (valid && f < F) ? x0[((long)b * F + f) * K + k] : 0.f;
  const float* xprow = xp + ((long)b * Hp) * K + k;
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
    p = fmaf(valid ? xprow[(long)h * K] : 0.f, t, p);
  }
  p += __shfl_xor(p, 16);
  p += __shfl_xor(p, 32);
  if (valid && hq == 0) pool[m] = p + bsum;
}

// Gprev[m,h] = dP[m] * sum_f x[m,f] wsum[h,f] (+ dPprev[m]);  dX[m,f] = dP[m] * sum_h x^{L-1}[m,h] wsum[h,f]
// layer1 (L == 1, x^{L-1} == x): both terms go to dX (the first one through a small LDS tile).
__global__ __launch_bounds__(256) void cin_last_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ xp,
                                                           const float* __restrict__ wsum, const float* __restrict__ dP, int ldp,
                                                           const float* __restrict__ dPprev, float* __restrict__ Gprev,
                                                           float* __restrict__ dX, int layer1, int M, int F, int K, int Hp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int FP4 = (F + 3) & ~3;
  float* ts = smem + Hp * FP4;  // [kLastRows][kLastFMax + 1], layer1 only
  stage_wsum(wsum, smem, Hp, F, FP4);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowl = wave * 16 + (lane & 15), hq = lane >> 4;
  const int m = blockIdx.x * kLastRows + rowl;
  const bool valid = m < M;
  const int b = valid ? m / K : 0, k = valid ? m - b * K : 0;
  float xr[kLastFMax], u[kLastFMax];
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) {
    xr[f] = (valid && f < F) ? x0[((long)b * F + f) * K + k] : 0.f;
    u[f] = 0.f;
  }
  const float dp = valid ? dP[(long)b * ldp + k] : 0.f;
  const float dpp = (valid && dPprev != nullptr) ? dPprev[(long)b * ldp + k] : 0.f;
  const float* xprow = xp + ((long)b * Hp) * K + k;
  for (int h = hq; h < Hp; h += 4) {
    const float4* wrow = reinterpret_cast<const float4*>(smem + h * FP4);
    const float xph = valid ? xprow[(long)h * K] : 0.f;
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
    else if (valid) Gprev[((long)b * Hp + h) * K + k] = fmaf(dp, t, dpp);
  }
  if (layer1) __syncthreads();
#pragma unroll
  for (int f = 0; f < kLastFMax; ++f) {
    if (f < F) {
      float v = u[f];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (layer1) v += ts[rowl * (kLastFMax + 1) + f];  // Hp == F: the x^{0} role of x
      if (valid && hq == 0) dX[((long)b * F + f) * K + k] = dp * v;
    }
  }
}

// dW[c,n] = v[c] for every n
__global__ __launch_bounds__(256) void cin_fill_rows_kernel(const float* __restrict__ v, float* __restrict__ dW, long C, int H) {
  const long total = C * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) dW[i] = v[i / H];
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

constexpr int kLastDwBlocks = 256;  // m-range splits of the last layer's weight-gradient partials

static size_t fwd_ws_bytes(const CinShape& s) {
  size_t t = 0;
  for (int l = 0; l < s.L; ++l) t += align_up((size_t)chunks_of(s.H[l]) * s.M() * sizeof(float), 256);
  t += align_up((size_t)s.Hp(s.L - 1) * s.F * sizeof(float), 256);  // wsum of the last layer
  return t;
}

constexpr int kHeadChunk = 64;   // samples per block in the head partial reductions
constexpr int kColChunk = 8;     // samples per block in the dbias (column-sum) partial reductions

static size_t bwd_ws_bytes(const CinShape& s) {
  size_t t = 0;
  const size_t LK = (size_t)s.L * s.K;
  t += align_up((size_t)s.B * LK * sizeof(float), 256);                                  // dP
  t += 2 * align_up((size_t)s.B * s.Hmax() * s.K * sizeof(float), 256);                  // G ping-pong
  size_t pmax = 0;
  for (int l = 0; l < s.L; ++l) pmax = std::max(pmax, (size_t)dw_splits(s, l) * s.Hp(l) * s.F * s.H[l]);
  t += align_up(pmax * sizeof(float), 256);                                              // dW partials
  const size_t nblk = (size_t)cdiv(std::max(1, s.B), kHeadChunk);
  const size_t ncol = (size_t)cdiv(std::max(1, s.B), kColChunk);
  t += align_up(std::max(ncol * s.Hmax(), nblk * (LK + 1)) * sizeof(float), 256);          // colsum / head partials
  const size_t cl = (size_t)s.Hp(s.L - 1) * s.F;
  t += 2 * align_up(cl * sizeof(float), 256);                                            // wsum, v of the last layer
  t += align_up((size_t)kLastDwBlocks * cl * sizeof(float), 256);                        // last-layer dW partials
  t += align_up((size_t)s.B * s.F * s.K * sizeof(float), 256);                           // x * dP_L
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
  const dim3 grid(cdiv(M, 128));
  const int xs_words = (2 * 128 * XS + 3) & ~3;
#define FIL_DZ(NH)                                                                                                  \
  {                                                                                                                 \
    const size_t sh = ((size_t)xs_words + 2 * (size_t)32 * (2 * NH + 4)) * sizeof(float);                           \
    allow_lds(cin_bwd_dz_kernel<NH>, sh);                                                                           \
    hipLaunchKernelGGL((cin_bwd_dz_kernel<NH>), grid, dim3(kCinThreads), sh, st, G, W, x0, xp, dPprev, ldp, Gprev, dX, \
                       accumulate, layer1, M, F, K, Hp, H);                                                         \
  }
  if (H <= 32) FIL_DZ(16)
  else if (H <= 64) FIL_DZ(32)
  else if (H <= 128) FIL_DZ(64)
  else FIL_DZ(128)
#undef FIL_DZ
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
  if (mode != 0 && mode != 1) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_fwd: mode %d (0 = fp32 MFMA + last-layer shortcut, 1 = fp32 MFMA, general kernels only)", mode);
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
  float* wsum_buf = nullptr;
  {
    Carver tail(workspace);  // the wsum slot sits after the pool partials (same order as fwd_ws_bytes)
    for (int l = 0; l < L; ++l) tail.take<float>((size_t)chunks_of(H[l]) * M);
    wsum_buf = tail.take<float>((size_t)s.Hp(L - 1) * F);
  }
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
    if (l == L - 1 && mode == 0) {
      // last layer: only its sum-pool is observable -> contract with wsum[c] = sum_n W[c,n]
      const int Hp = s.Hp(l);
      float* wsum = wsum_buf;
      const size_t sh = (size_t)Hp * ((F + 3) & ~3) * sizeof(float);
      ProfScope ps("cin_last_fwd", st, 2.0 * (double)M * Hp * F);
      hipLaunchKernelGGL(cin_wsum_kernel, dim3(cdiv(Hp * F, 8)), dim3(256), 0, st, W[l], wsum, Hp * F, H[l]);
      allow_lds(cin_last_fwd_kernel, sh);
      hipLaunchKernelGGL(cin_last_fwd_kernel, dim3(cdiv(M, kLastRows)), dim3(256), sh, st, x, xp, wsum, bias[l], part, M, F, K, Hp, H[l]);
      pa.chunks[l] = 1;
    } else {
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
  if (mode != 0 && mode != 1) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_bwd: mode %d (0 = fp32 MFMA + last-layer shortcut, 1 = fp32 MFMA, general kernels only)", mode);
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
  const int ncol = cdiv(B, kColChunk);
  float* small = ws.take<float>(std::max((size_t)ncol * s.Hmax(), (size_t)nblk * (LK + 1)));
  const size_t cl = (size_t)s.Hp(L - 1) * F;
  float* wsum = ws.take<float>(cl);
  float* vlast = ws.take<float>(cl);
  float* lastpart = ws.take<float>((size_t)kLastDwBlocks * cl);
  float* ybuf = ws.take<float>((size_t)B * F * K);

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
    hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv((int)LK + 1, 64)), dim3(256), 0, st, small, tmp, (long)(LK + 1), nblk);
    FIL_CHECK_LAUNCH();
    (void)hipMemcpyAsync(ddense_w, tmp, LK * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(ddense_b, tmp + LK, sizeof(float), hipMemcpyDeviceToDevice, st);
  }

  int cur = 0;
  int ltop = L - 1;  // first layer handled by the general kernels
  if (mode == 0) {
    // ---- last layer through the pooled-weights shortcut (see cin_last_* kernels)
    const int l = L - 1;
    FIL_CHECK_ARG(W[l] && dW[l] && dbias[l]);
    const int Hp = s.Hp(l), Hl = H[l];
    const float* xp = l == 0 ? x : maps[l - 1];
    const float* dPl = dPsrc + (size_t)l * K;
    const float* dPprev = l > 0 ? dPsrc + (size_t)(l - 1) * K : nullptr;
    const size_t shw = (size_t)Hp * ((F + 3) & ~3) * sizeof(float);
    ProfScope ps("cin_last_bwd", st, 6.0 * (double)M * Hp * F);
    hipLaunchKernelGGL(cin_wsum_kernel, dim3(cdiv(Hp * F, 8)), dim3(256), 0, st, W[l], wsum, Hp * F, Hl);
    hipLaunchKernelGGL(cin_slice_sum_kernel, dim3(nblk), dim3(256), 0, st, dPl, (int)LK, small, B, K, kHeadChunk);
    hipLaunchKernelGGL(cin_fill_sum_kernel, dim3(1), dim3(256), 0, st, small, nblk, dbias[l], Hl);
    // dW_L[c,:] = v[c],  v[h,f] = sum_m x^{L-1}[m,h] * (x[m,f] dP[m]): the MFMA weight-gradient kernel with a
    // single-field "x" of ones (F' = 1, so c = h) and G = x * dP as its [B, F, K] right-hand side
    {
      const long tot = (long)B * F * K;
      hipLaunchKernelGGL(cin_scale_rows_kernel, dim3((int)std::min<long>((tot + 255) / 256, 2048)), dim3(256), 0, st, x, dPl, (int)LK,
                         ybuf, B, F, K);
      const int nb = launch_dw(st, ybuf, nullptr, xp, lastpart, B, /*F=*/1, K, Hp, /*H=*/F, std::min(kLastDwBlocks, B));
      hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv((int)cl, 64)), dim3(256), 0, st, lastpart, vlast, (long)cl, nb);
    }
    hipLaunchKernelGGL(cin_fill_rows_kernel, dim3((int)std::min<long>(((long)cl * Hl + 255) / 256, 2048)), dim3(256), 0, st, vlast,
                       dW[l], (long)cl, Hl);
    // G^{L-1} and dX
    const size_t shb = shw + (l == 0 ? (size_t)kLastRows * (kLastFMax + 1) * sizeof(float) : 0);
    allow_lds(cin_last_bwd_kernel, shb);
    hipLaunchKernelGGL(cin_last_bwd_kernel, dim3(cdiv(M, kLastRows)), dim3(256), shb, st, x, xp, wsum, dPl, (int)LK, dPprev,
                       l > 0 ? Gbuf[cur] : nullptr, dx, /*layer1=*/l == 0, M, F, K, Hp);
    FIL_CHECK_LAUNCH();
    ltop = L - 2;
  } else {
    // ---- top layer gradient: broadcast of its pooled gradient
    const long total = (long)B * H[L - 1] * K;
    const int grid = (int)std::min<long>((total + 255) / 256, 4096);
    ProfScope ps("cin_bcast_g", st, (double)total * sizeof(float));
    hipLaunchKernelGGL(cin_bcast_kernel, dim3(grid), dim3(256), 0, st, dPsrc + (size_t)(L - 1) * K, (int)LK, Gbuf[cur], B, H[L - 1], K);
    FIL_CHECK_LAUNCH();
  }
  for (int l = ltop; l >= 0; --l) {
    FIL_CHECK_ARG(W[l] && dW[l] && dbias[l]);
    const int Hp = s.Hp(l), Hl = H[l];
    const float* xp = l == 0 ? x : maps[l - 1];
    const float* G = Gbuf[cur];
    // dbias
    {
      ProfScope ps("cin_dbias", st, (double)B * Hl * K * sizeof(float));
      hipLaunchKernelGGL(cin_colsum_kernel, dim3(ncol), dim3(256), (size_t)Hl * K * sizeof(float), st, G, small, B, Hl, K, kColChunk);
      hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv(Hl, 64)), dim3(256), 0, st, small, dbias[l], (long)Hl, ncol);
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
      hipLaunchKernelGGL(cin_reduce_kernel, dim3((int)((nW + 63) / 64)), dim3(256), 0, st, part, dW[l], nW, parts);
    }
    FIL_CHECK_LAUNCH();
    // dZ -> G^{l-1}, dX
    const float* dPprev = l > 0 ? dPsrc + (size_t)(l - 1) * K : nullptr;
    {
      ProfScope ps(kDzNames[l], st, gemm_flops(M, Hp, F, Hl));
      launch_dz(st, G, W[l], x, xp, dPprev, (int)LK, l > 0 ? Gbuf[cur ^ 1] : nullptr, dx, /*accumulate=*/!(mode == 1 && l == L - 1),
                /*layer1=*/l == 0, M, F, K, Hp, Hl);
    }
    FIL_CHECK_LAUNCH();
    cur ^= 1;
  }
  return FIL_OK;
}
