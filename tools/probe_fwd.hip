// Micro-benchmark harness for the CIN forward implicit GEMM (not part of the library): kernel-structure variants
// timed with hipEvents on random data and spot-checked against a host fp64 evaluation.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_fwd.hip -o /tmp/probe_fwd && /tmp/probe_fwd
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }

constexpr int F = 39, K = 16, J = 20, FPAD = 40, XS = 41, NW = 128;

// ---------------------------------------------------------------------------------------------------------------
// Variant A: 4 waves x 32 rows, x row fragment in registers (J static), W slab in "lane-major" layout
// Ws[f][r][nb] so that one ds_read_b128 fetches the 4 B operands of a step; ping-pong operand registers.
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void fwd_A(const float* __restrict__ x0, const float* __restrict__ xp,
                                                             const float* __restrict__ W, float* __restrict__ xout, int M, int Hp,
                                                             int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;  // [2][FPAD][32][4]
  constexpr int slab = FPAD * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * 128;
  const int m = row0 + wave * 32 + r;
  const int b = m / K, k = m - b * K;
  float xr[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int f = 2 * j + half;
    xr[j] = f < F ? x0[((long)b * F + f) * K + k] : 0.f;
  }
  constexpr int NSL = slab / 256;  // 20
  float pw[NSL];
  auto issue_slab = [&](int h) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * 256;
      const int f = idx >> 7, n = idx & 127;
      pw[u] = f < F ? W[((long)h * F + f) * H + n] : 0.f;
    }
  };
  auto commit_slab = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * 256;
      const int f = idx >> 7, n = idx & 127;
      dst[f * 128 + (n & 31) * 4 + (n >> 5)] = pw[u];
    }
  };
  issue_slab(0);
  commit_slab(Ws);
  f32x16 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  const float* xprow = xp + ((long)b * Hp) * K + k;
  float xpv = xprow[0], xpn = 0.f;
  __syncthreads();
  for (int h = 0; h < Hp; ++h) {
    const int buf = h & 1;
    const bool more = h + 1 < Hp;
    settle(xpv);
    if (more) {
      xpn = xprow[(long)(h + 1) * K];
      issue_slab(h + 1);
    }
    const float4* wrow = reinterpret_cast<const float4*>(Ws + buf * slab + half * 128 + r * 4);  // + (2j)*32 float4
    float4 wa = wrow[0], wbq;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      wbq = wrow[(2 * (j + 1)) * 32];
      {
        const float a = xpv * xr[j];
        acc[0] = mfma32(a, wa.x, acc[0]);
        acc[1] = mfma32(a, wa.y, acc[1]);
        acc[2] = mfma32(a, wa.z, acc[2]);
        acc[3] = mfma32(a, wa.w, acc[3]);
      }
      if (j + 2 < J) wa = wrow[(2 * (j + 2)) * 32];
      {
        const float a = xpv * xr[j + 1];
        acc[0] = mfma32(a, wbq.x, acc[0]);
        acc[1] = mfma32(a, wbq.y, acc[1]);
        acc[2] = mfma32(a, wbq.z, acc[2]);
        acc[3] = mfma32(a, wbq.w, acc[3]);
      }
    }
    if (more) {
      commit_slab(Ws + (buf ^ 1) * slab);
      xpv = xpn;
    }
    __syncthreads();
  }
  const int wrow0 = row0 + wave * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mm = wrow0 + 8 * q + 4 * half;
    const int bb = mm / K, kk = mm - bb * K;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = nb * 32 + r;
      *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
          make_float4(acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Variant D: 4 waves x 64 rows (two 32-row M blocks per wave, 8 accumulators), one wave per SIMD.
__global__ __launch_bounds__(256, 1) void fwd_D(const float* __restrict__ x0, const float* __restrict__ xp,
                                                const float* __restrict__ W, float* __restrict__ xout, int M, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  constexpr int slab = FPAD * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int row0 = blockIdx.x * 256;
  int bq[2], kq[2];
  float xr[2][J];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = row0 + wave * 64 + mb * 32 + r;
    bq[mb] = m / K;
    kq[mb] = m - bq[mb] * K;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int f = 2 * j + half;
      xr[mb][j] = f < F ? x0[((long)bq[mb] * F + f) * K + kq[mb]] : 0.f;
    }
  }
  constexpr int NSL = slab / 256;
  float pw[NSL];
  auto issue_slab = [&](int h) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * 256;
      const int f = idx >> 7, n = idx & 127;
      pw[u] = f < F ? W[((long)h * F + f) * H + n] : 0.f;
    }
  };
  auto commit_slab = [&](float* dst) {
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * 256;
      const int f = idx >> 7, n = idx & 127;
      dst[f * 128 + (n & 31) * 4 + (n >> 5)] = pw[u];
    }
  };
  issue_slab(0);
  commit_slab(Ws);
  f32x16 acc[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  const float* xprow0 = xp + ((long)bq[0] * Hp) * K + kq[0];
  const float* xprow1 = xp + ((long)bq[1] * Hp) * K + kq[1];
  float xpv0 = xprow0[0], xpv1 = xprow1[0], xpn0 = 0.f, xpn1 = 0.f;
  __syncthreads();
  for (int h = 0; h < Hp; ++h) {
    const int buf = h & 1;
    const bool more = h + 1 < Hp;
    settle(xpv0);
    settle(xpv1);
    if (more) {
      xpn0 = xprow0[(long)(h + 1) * K];
      xpn1 = xprow1[(long)(h + 1) * K];
      issue_slab(h + 1);
    }
    const float4* wrow = reinterpret_cast<const float4*>(Ws + buf * slab + half * 128 + r * 4);
    float4 wa = wrow[0], wbq;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      wbq = wrow[(2 * (j + 1)) * 32];
      {
        const float a0 = xpv0 * xr[0][j], a1 = xpv1 * xr[1][j];
        acc[0][0] = mfma32(a0, wa.x, acc[0][0]);
        acc[1][0] = mfma32(a1, wa.x, acc[1][0]);
        acc[0][1] = mfma32(a0, wa.y, acc[0][1]);
        acc[1][1] = mfma32(a1, wa.y, acc[1][1]);
        acc[0][2] = mfma32(a0, wa.z, acc[0][2]);
        acc[1][2] = mfma32(a1, wa.z, acc[1][2]);
        acc[0][3] = mfma32(a0, wa.w, acc[0][3]);
        acc[1][3] = mfma32(a1, wa.w, acc[1][3]);
      }
      if (j + 2 < J) wa = wrow[(2 * (j + 2)) * 32];
      {
        const float a0 = xpv0 * xr[0][j + 1], a1 = xpv1 * xr[1][j + 1];
        acc[0][0] = mfma32(a0, wbq.x, acc[0][0]);
        acc[1][0] = mfma32(a1, wbq.x, acc[1][0]);
        acc[0][1] = mfma32(a0, wbq.y, acc[0][1]);
        acc[1][1] = mfma32(a1, wbq.y, acc[1][1]);
        acc[0][2] = mfma32(a0, wbq.z, acc[0][2]);
        acc[1][2] = mfma32(a1, wbq.z, acc[1][2]);
        acc[0][3] = mfma32(a0, wbq.w, acc[0][3]);
        acc[1][3] = mfma32(a1, wbq.w, acc[1][3]);
      }
    }
    if (more) {
      commit_slab(Ws + (buf ^ 1) * slab);
      xpv0 = xpn0;
      xpv1 = xpn1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int wrow0 = row0 + wave * 64 + mb * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mm = wrow0 + 8 * q + 4 * half;
      const int bb = mm / K, kk = mm - bb * K;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = nb * 32 + r;
        *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
            make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Variant E: no LDS, no barriers.  W is pre-arranged lane-major in global memory: Wt[h][FPAD][32][4]
// (Wt[(h*FPAD + f)*128 + r*4 + nb] = W[(h*F+f)*H + nb*32 + r], zero rows for f >= F), so the 4 B operands of a step
// are ONE coalesced global_load_dwordx4 per lane; every wave streams W from L2 on its own with a DEPTH-step
// register prefetch queue.  MB = 32-row blocks per wave (1 or 2).
__global__ void transpose_w(const float* __restrict__ W, float* __restrict__ Wt, int Hp, int H) {
  const long total = (long)Hp * FPAD * 128;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nb = i & 3, r = (i >> 2) & 31;
    const long cf = i >> 7;
    const int f = cf % FPAD, h = cf / FPAD;
    const int n = nb * 32 + r;
    Wt[i] = (f < F && n < H) ? W[((long)h * F + f) * H + n] : 0.f;
  }
}

template <int MB, int DEPTH, int WPS>
__global__ __launch_bounds__(256, WPS) void fwd_E(const float* __restrict__ x0, const float* __restrict__ xp,
                                                  const float* __restrict__ Wt, float* __restrict__ xout, int M, int Hp, int H) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * (32 * MB);
  int bq[MB], kq[MB];
  float xr[MB][J];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wrow0 + mb * 32 + r;
    bq[mb] = m / K;
    kq[mb] = m - bq[mb] * K;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int f = 2 * j + half;
      xr[mb][j] = f < F ? x0[((long)bq[mb] * F + f) * K + kq[mb]] : 0.f;
    }
  }
  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  // W stream: step s = h*J + j reads row (h*FPAD + 2j + half): rows advance by 2 per step inside an h
  const float4* wbase = reinterpret_cast<const float4*>(Wt) + (half * 32 + r);
  float4 q[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) q[d] = wbase[(long)(2 * d) * 32];  // steps 0..DEPTH-1 of h = 0 (DEPTH <= J)
  const float* xprow[MB];
  float xpv[MB], xpn[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    xprow[mb] = xp + ((long)bq[mb] * Hp) * K + kq[mb];
    xpv[mb] = xprow[mb][0];
    xpn[mb] = 0.f;
  }
  for (int h = 0; h < Hp; ++h) {
    const bool more = h + 1 < Hp;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      if (more) xpn[mb] = xprow[mb][(long)(h + 1) * K];
    const float4* wh = wbase + (long)h * FPAD * 32;
    const float4* whn = wbase + (long)(more ? h + 1 : h) * FPAD * 32;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float4 w = q[j % DEPTH];
      // refill this queue slot with the operand of step j + DEPTH (possibly in the next h)
      const int jn = j + DEPTH;
      q[j % DEPTH] = jn < J ? wh[(long)(2 * jn) * 32] : whn[(long)(2 * (jn - J)) * 32];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float a = xpv[mb] * xr[mb][j];
        acc[mb][0] = mfma32(a, w.x, acc[mb][0]);
        acc[mb][1] = mfma32(a, w.y, acc[mb][1]);
        acc[mb][2] = mfma32(a, w.z, acc[mb][2]);
        acc[mb][3] = mfma32(a, w.w, acc[mb][3]);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the refill load of step j+DEPTH here: the scheduler otherwise sinks it to its use
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xpv[mb] = xpn[mb];
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int mm = wrow0 + mb * 32 + 8 * q4 + 4 * half;
      const int bb = mm / K, kk = mm - bb * K;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = nb * 32 + r;
        *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
            make_float4(acc[mb][nb][4 * q4], acc[mb][nb][4 * q4 + 1], acc[mb][nb][4 * q4 + 2], acc[mb][nb][4 * q4 + 3]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Variant F: generic (runtime F): flat step stream s = h*J + j over the lane-major W (row 2s+half), DEPTH-deep
// register queue refilled in place, x operand from a wave-private LDS tile, x^{l-1} value switched when j wraps.
template <int MB, int DEPTH>
__global__ __launch_bounds__(256, 1) void fwd_F(const float* __restrict__ x0, const float* __restrict__ xp,
                                                const float* __restrict__ Wt, float* __restrict__ xout, int M, int Fr, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Jr = (Fr + 1) >> 1, XSr = 2 * Jr + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * (32 * MB);
  float* xs = smem + wave * (32 * MB) * XSr;  // wave-private [32*MB][XS]
  int bq[MB], kq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wrow0 + mb * 32 + r;
    bq[mb] = m / K;
    kq[mb] = m - bq[mb] * K;
    for (int f = half; f < 2 * Jr; f += 2)
      xs[(mb * 32 + r) * XSr + f] = f < Fr ? x0[((long)bq[mb] * Fr + f) * K + kq[mb]] : 0.f;
  }
  __syncthreads();
  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  const float4* wp = reinterpret_cast<const float4*>(Wt) + (half * 32 + r);  // step s: wp[s * 64]
  const int S = Hp * Jr;
  const int Spad = (S + DEPTH - 1) / DEPTH * DEPTH;  // Wt is allocated (and zero filled) up to Spad + DEPTH steps
  float4 q[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) q[d] = wp[(long)d * 64];
  const float* xprow[MB];
  float xpv[MB], xpn[MB], xa[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    xprow[mb] = xp + ((long)bq[mb] * Hp) * K + kq[mb];
    xpv[mb] = xprow[mb][0];
    xpn[mb] = Hp > 1 ? xprow[mb][K] : 0.f;
    xa[mb] = xs[(mb * 32 + r) * XSr + half];
  }
  int j = 0, h = 0;
  for (int s0 = 0; s0 < Spad; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const float4 w = q[d];
      q[d] = wp[(long)(s0 + d + DEPTH) * 64];
      float a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = xpv[mb] * xa[mb];
      // advance (h, j); prefetch the next step's x operand; switch x^{l-1} when j wraps
      if (++j == Jr) {
        j = 0;
        ++h;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          xpv[mb] = h < Hp ? xpn[mb] : 0.f;
          xpn[mb] = h + 1 < Hp ? xprow[mb][(long)(h + 1) * K] : 0.f;
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xa[mb] = xs[(mb * 32 + r) * XSr + 2 * j + half];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][0] = mfma32(a[mb], w.x, acc[mb][0]);
        acc[mb][1] = mfma32(a[mb], w.y, acc[mb][1]);
        acc[mb][2] = mfma32(a[mb], w.z, acc[mb][2]);
        acc[mb][3] = mfma32(a[mb], w.w, acc[mb][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int mm = wrow0 + mb * 32 + 8 * q4 + 4 * half;
      const int bb = mm / K, kk = mm - bb * K;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = nb * 32 + r;
        *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
            make_float4(acc[mb][nb][4 * q4], acc[mb][nb][4 * q4 + 1], acc[mb][nb][4 * q4 + 2], acc[mb][nb][4 * q4 + 3]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Variant G: generic (runtime F), nested dynamic loops with straight-line bodies: per h, J is padded to a
// multiple of U and processed in groups of U steps (U = queue depth = unroll); W rows per h = 2*Jpad (lane-major,
// zero rows for padding), x operand from a wave-private LDS tile (zero padded), no branches inside a group.
template <int MB, int U>
__global__ __launch_bounds__(256, 1) void fwd_G(const float* __restrict__ x0, const float* __restrict__ xp,
                                                const float* __restrict__ Wt, float* __restrict__ xout, int M, int Fr, int Hp, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Jr = (Fr + 1) >> 1;
  const int G = (Jr + U - 1) / U, Jp = G * U, XSr = 2 * Jp + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int wrow0 = (blockIdx.x * 4 + wave) * (32 * MB);
  float* xs = smem + wave * (32 * MB) * XSr;
  int bq[MB], kq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wrow0 + mb * 32 + r;
    bq[mb] = m / K;
    kq[mb] = m - bq[mb] * K;
    for (int f = half; f < 2 * Jp; f += 2)
      xs[(mb * 32 + r) * XSr + f] = f < Fr ? x0[((long)bq[mb] * Fr + f) * K + kq[mb]] : 0.f;
  }
  __syncthreads();
  f32x16 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  const float4* wp = reinterpret_cast<const float4*>(Wt) + (half * 32 + r);  // step s (flat, padded): wp[s * 64]
  float4 q[U];
#pragma unroll
  for (int d = 0; d < U; ++d) q[d] = wp[(long)d * 64];
  wp += (long)U * 64;  // points at the refill source of the step being consumed
  const float* xprow[MB];
  const float* xsrow[MB];
  float xpv[MB], xpn[MB], xa[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    xprow[mb] = xp + ((long)bq[mb] * Hp) * K + kq[mb];
    xsrow[mb] = xs + (mb * 32 + r) * XSr + half;
    xpv[mb] = xprow[mb][0];
    xa[mb] = xsrow[mb][0];
  }
  for (int h = 0; h < Hp; ++h) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xpn[mb] = h + 1 < Hp ? xprow[mb][(long)(h + 1) * K] : 0.f;
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int d = 0; d < U; ++d) {
        const float4 w = q[d];
        q[d] = wp[(long)d * 64];   // the stream is allocated (zero filled) U steps past the end
        float a[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = xpv[mb] * xa[mb];
        // next step's x operand (wraps to column 0 at the end of the h)
        const int jn = (g * U + d + 1 == Jp) ? 0 : g * U + d + 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) xa[mb] = xsrow[mb][2 * jn];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb][0] = mfma32(a[mb], w.x, acc[mb][0]);
          acc[mb][1] = mfma32(a[mb], w.y, acc[mb][1]);
          acc[mb][2] = mfma32(a[mb], w.z, acc[mb][2]);
          acc[mb][3] = mfma32(a[mb], w.w, acc[mb][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      wp += (long)U * 64;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) xpv[mb] = xpn[mb];
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int mm = wrow0 + mb * 32 + 8 * q4 + 4 * half;
      const int bb = mm / K, kk = mm - bb * K;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = nb * 32 + r;
        *reinterpret_cast<float4*>(xout + ((long)bb * H + n) * K + kk) =
            make_float4(acc[mb][nb][4 * q4], acc[mb][nb][4 * q4 + 1], acc[mb][nb][4 * q4 + 2], acc[mb][nb][4 * q4 + 3]);
      }
    }
  }
}

// pure MFMA ceiling: same MFMA count per wave as the real kernel, operands in registers
template <int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_only(float* out, int iters) {
  f32x16 acc[4];
  for (int nb = 0; nb < 4; ++nb) for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = mfma32(a, b, acc[0]);
      acc[1] = mfma32(a, b, acc[1]);
      acc[2] = mfma32(a, b, acc[2]);
      acc[3] = mfma32(a, b, acc[3]);
    }
  }
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) for (int i = 0; i < 16; ++i) s += acc[nb][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int B = 4096, Hp = 128, H = 128, M = B * K;
  std::vector<float> hx((size_t)B * F * K), hxp((size_t)B * Hp * K), hW((size_t)Hp * F * H);
  srand(1);
  auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hxp) v = rnd();
  for (auto& v : hW) v = rnd() * 0.05f;
  float *dx, *dxp, *dW, *dout;
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dxp, hxp.size() * 4)); CHECK(hipMalloc(&dW, hW.size() * 4));
  CHECK(hipMalloc(&dout, (size_t)B * H * K * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dxp, hxp.data(), hxp.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double flops = 2.0 * M * Hp * F * H;
  std::vector<float> hout((size_t)B * H * K);
  auto check = [&](const char* name) {
    CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int t = 0; t < 64; ++t) {
      const int b = (t * 977) % B, k = (t * 7) % K, n = (t * 37) % H;
      double ref = 0;
      for (int h = 0; h < Hp; ++h)
        for (int f = 0; f < F; ++f)
          ref += (double)hxp[((size_t)b * Hp + h) * K + k] * hx[((size_t)b * F + f) * K + k] * hW[((size_t)h * F + f) * H + n];
      const double got = hout[((size_t)b * H + n) * K + k];
      maxerr = fmax(maxerr, fabs(got - ref));
      maxref = fmax(maxref, fabs(ref));
    }
    printf("  %s check: max err %.3e (max |ref| %.3e)\n", name, maxerr, maxref);
  };
  auto timeit = [&](const char* name, auto launch, double fl) {
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
  };
  {
    const size_t sh = 2 * FPAD * NW * 4;
    CHECK(hipFuncSetAttribute((const void*)fwd_A<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    CHECK(hipFuncSetAttribute((const void*)fwd_A<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    CHECK(hipFuncSetAttribute((const void*)fwd_D, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    CHECK(hipMemset(dout, 0, hout.size() * 4));
    timeit("A: 32 rows/wave, 2 waves/SIMD", [&] { hipLaunchKernelGGL(fwd_A<2>, dim3(M / 128), dim3(256), sh, 0, dx, dxp, dW, dout, M, Hp, H); }, flops);
    check("A");
    CHECK(hipMemset(dout, 0, hout.size() * 4));
    timeit("D: 64 rows/wave, 1 wave/SIMD", [&] { hipLaunchKernelGGL(fwd_D, dim3(M / 256), dim3(256), sh, 0, dx, dxp, dW, dout, M, Hp, H); }, flops);
    check("D");
  }
  {
    float* dWt;
    CHECK(hipMalloc(&dWt, (size_t)Hp * FPAD * 128 * 4));
    hipLaunchKernelGGL(transpose_w, dim3(512), dim3(256), 0, 0, dW, dWt, Hp, H);
    CHECK(hipMemset(dout, 0, hout.size() * 4));
    timeit("E: MB=2 D=4 1w/SIMD", [&] { hipLaunchKernelGGL((fwd_E<2, 4, 1>), dim3(M / 256), dim3(256), 0, 0, dx, dxp, dWt, dout, M, Hp, H); }, flops);
    check("E MB=2 D=4");
    timeit("E: MB=2 D=5 1w/SIMD", [&] { hipLaunchKernelGGL((fwd_E<2, 5, 1>), dim3(M / 256), dim3(256), 0, 0, dx, dxp, dWt, dout, M, Hp, H); }, flops);
    timeit("E: MB=2 D=10 1w/SIMD", [&] { hipLaunchKernelGGL((fwd_E<2, 10, 1>), dim3(M / 256), dim3(256), 0, 0, dx, dxp, dWt, dout, M, Hp, H); }, flops);
    CHECK(hipMemset(dout, 0, hout.size() * 4));
    timeit("E: MB=1 D=4 2w/SIMD", [&] { hipLaunchKernelGGL((fwd_E<1, 4, 2>), dim3(M / 128), dim3(256), 0, 0, dx, dxp, dWt, dout, M, Hp, H); }, flops);
    check("E MB=1 D=4");
    timeit("E: MB=1 D=10 2w/SIMD", [&] { hipLaunchKernelGGL((fwd_E<1, 10, 2>), dim3(M / 128), dim3(256), 0, 0, dx, dxp, dWt, dout, M, Hp, H); }, flops);
    {
      // F variant: Wt needs Spad + DEPTH rows-pairs; allocate generously and zero the tail
      float* dWt2;
      const size_t rows = (size_t)Hp * FPAD + 64;
      CHECK(hipMalloc(&dWt2, rows * 128 * 4));
      CHECK(hipMemset(dWt2, 0, rows * 128 * 4));
      hipLaunchKernelGGL(transpose_w, dim3(512), dim3(256), 0, 0, dW, dWt2, Hp, H);
      const size_t shF = 4 * 64 * XS * 4;
      CHECK(hipMemset(dout, 0, hout.size() * 4));
      timeit("F: MB=2 D=4 generic", [&] { hipLaunchKernelGGL((fwd_F<2, 4>), dim3(M / 256), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      check("F MB=2 D=4");
      timeit("F: MB=2 D=8 generic", [&] { hipLaunchKernelGGL((fwd_F<2, 8>), dim3(M / 256), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      check("F MB=2 D=8");
      timeit("F: MB=2 D=12 generic", [&] { hipLaunchKernelGGL((fwd_F<2, 12>), dim3(M / 256), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      timeit("G: MB=2 U=10 generic", [&] { hipLaunchKernelGGL((fwd_G<2, 10>), dim3(M / 256), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      check("G MB=2 U=10");
      timeit("G: MB=2 U=5 generic", [&] { hipLaunchKernelGGL((fwd_G<2, 5>), dim3(M / 256), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      check("G MB=2 U=5");
      timeit("G: MB=1 U=10 generic", [&] { hipLaunchKernelGGL((fwd_G<1, 10>), dim3(M / 128), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
      timeit("F: MB=1 D=8 generic", [&] { hipLaunchKernelGGL((fwd_F<1, 8>), dim3(M / 128), dim3(256), shF, 0, dx, dxp, dWt2, dout, M, F, Hp, H); }, flops);
    }
    timeit("transpose_w", [&] { hipLaunchKernelGGL(transpose_w, dim3(512), dim3(256), 0, 0, dW, dWt, Hp, H); }, 0);
  }
  {
    const int iters = 320;  // 320*32 = 10240 MFMAs per wave, as the layer-2 forward
    timeit("mfma_only 2048 waves (2/SIMD)", [&] { hipLaunchKernelGGL(mfma_only<2>, dim3(512), dim3(256), 0, 0, dout, iters); }, 512.0 * 4 * 10240 * 4096);
    timeit("mfma_only 1024 waves (1/SIMD)", [&] { hipLaunchKernelGGL(mfma_only<1>, dim3(256), dim3(256), 0, 0, dout, 2 * iters); }, 256.0 * 4 * 20480 * 4096);
  }
  return 0;
}
