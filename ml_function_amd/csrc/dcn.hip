// A2  DCN cross network, all L layers fused, for gfx950.
//
// Replaces CrossLayer.call of the reference (interactive_layer.py:275-282): per layer one [B,1,D]x[D,1]
// K.dot, one K.batch_dot and two adds, each streaming [B,D] through memory.  Here one wavefront owns one
// sample: its D floats live in the wave's registers (NPL per lane, 16-byte coalesced loads), the L dot
// products are wave reductions, w/b are staged once per workgroup in LDS, and x is read once and y
// written once (fwd: 2*D*4 bytes per sample; bwd: read x,g, write dx: 3*D*4 bytes per sample).
// Backward recomputes x_l from x0 and the saved dots s_l (bit-identical to forward), accumulates dw/db per
// lane in registers across the wave's samples, and reduces them in a fixed order:
// registers -> LDS (waves of a block) -> partial[block] -> dcn_reduce_kernel.  No atomics.
#include "common.h"
#include <cstdlib>

namespace fil {

constexpr int kDcnThreads = 256;  // forward: 4 waves per workgroup (1 per SIMD), one sample per wave at a time
constexpr int kDcnWaves = kDcnThreads / kWave;
// backward: 8 waves per workgroup = two per SIMD (its ~230 registers per lane allow exactly that), so a second wave covers
// the first one's load latency; the dw/db accumulators of the 8 waves fold through LDS into ONE partial per workgroup, so
// doubling the resident waves does not double the partial traffic (round 1 ran 4 waves per CU: 0.56 of peak at B = 131 k)
constexpr int kDcnBwdWaves = 8;
constexpr int kDcnBwdThreads = kDcnBwdWaves * kWave;
constexpr int kDcnMaxLc = 6;   // L limit of the closed-form backward (== kDcnMaxL below)

// Lane-owned element layout: chunk c = lane + 64*u (u < NPL/VEC) covers elements [c*VEC, c*VEC+VEC).
template <int NPL, int VEC>
__device__ __forceinline__ void row_load(const float* __restrict__ p, int D, int lane, float (&v)[NPL]) {
#pragma unroll
  for (int u = 0; u < NPL / VEC; ++u) {
    const int d = (lane + kWave * u) * VEC;
    if constexpr (VEC == 4) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d < D) t = *reinterpret_cast<const float4*>(p + d);
      v[u * 4 + 0] = t.x; v[u * 4 + 1] = t.y; v[u * 4 + 2] = t.z; v[u * 4 + 3] = t.w;
    } else {
      v[u] = d < D ? p[d] : 0.f;
    }
  }
}

template <int NPL, int VEC>
__device__ __forceinline__ void row_store(float* __restrict__ p, int D, int lane, const float (&v)[NPL]) {
#pragma unroll
  for (int u = 0; u < NPL / VEC; ++u) {
    const int d = (lane + kWave * u) * VEC;
    if (d < D) {
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(p + d) = make_float4(v[u * 4], v[u * 4 + 1], v[u * 4 + 2], v[u * 4 + 3]);
      else p[d] = v[u];
    }
  }
}

// Rows of the [B,D] tensors through raw buffers (one descriptor per row: lanes past D read 0 / store nothing, no branches,
// so a row's loads issue back to back behind one wait).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dcn_rsrc(const float* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
typedef float dcn_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int dcn_u4 __attribute__((ext_vector_type(4)));
template <int NPL, int VEC>
__device__ __forceinline__ void row_load_buf(__amdgpu_buffer_rsrc_t r, int lane, float (&v)[NPL]) {
#pragma unroll
  for (int u = 0; u < NPL / VEC; ++u) {
    const int off = (lane + kWave * u) * VEC * 4;
    if constexpr (VEC == 4) {
      const dcn_f4 t = __builtin_bit_cast(dcn_f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
      v[u * 4 + 0] = t[0]; v[u * 4 + 1] = t[1]; v[u * 4 + 2] = t[2]; v[u * 4 + 3] = t[3];
    } else {
      v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
  }
}
template <int NPL, int VEC>
__device__ __forceinline__ void row_store_buf(__amdgpu_buffer_rsrc_t r, int lane, const float (&v)[NPL]) {
#pragma unroll
  for (int u = 0; u < NPL / VEC; ++u) {
    const int off = (lane + kWave * u) * VEC * 4;
    if constexpr (VEC == 4) {
      const dcn_f4 t = {v[u * 4], v[u * 4 + 1], v[u * 4 + 2], v[u * 4 + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dcn_u4, t), r, off, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v[u]), r, off, 0, 0);
    }
  }
}

// wave total as a SCALAR: DPP adds inside each 32-lane half (VALU only, no LDS crossbar round trips like the ds_bpermute
// behind __shfl_xor), then the two half totals are read into scalar registers
__device__ __forceinline__ float wave_total(float v) {
  v = half_wave_sum_hi(v);
  const int bits = __builtin_bit_cast(int, v);   // (the builtin is typed int: a float argument would be VALUE-converted)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 31)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, 63));
}

// Parameter row l (w or b) from the LDS copy staged at kernel start.
template <int NPL, int VEC>
__device__ __forceinline__ void param_load(const float* p, int D, int lane, float (&v)[NPL]) {
  row_load<NPL, VEC>(p, D, lane, v);
}

template <int NPL, int VEC>
__global__ __launch_bounds__(kDcnThreads) void dcn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              float* __restrict__ s, int B, int D, int L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < L * D; i += kDcnThreads) {
    smem[i] = w[i];
    smem[L * D + i] = b[i];
  }
  __syncthreads();
  const float* ws = smem;
  const float* bs = smem + L * D;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: descriptors built from it stay in SGPRs, no waterfall loops)
  for (int n = blockIdx.x * kDcnWaves + wave; n < B; n += gridDim.x * kDcnWaves) {
    float x0[NPL], xl[NPL];
    row_load<NPL, VEC>(x + (long)n * D, D, lane, x0);
#pragma unroll
    for (int i = 0; i < NPL; ++i) xl[i] = x0[i];
    for (int l = 0; l < L; ++l) {
      float wl[NPL], bl[NPL];
      param_load<NPL, VEC>(ws + l * D, D, lane, wl);
      param_load<NPL, VEC>(bs + l * D, D, lane, bl);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) dot = fmaf(xl[i], wl[i], dot);
      dot = wave_sum(dot);
      if (lane == 0) s[(long)n * L + l] = dot;
#pragma unroll
      for (int i = 0; i < NPL; ++i) xl[i] = fmaf(x0[i], dot, xl[i]) + bl[i];
    }
    row_store<NPL, VEC>(y + (long)n * D, D, lane, xl);
  }
}

// Backward in closed form.  With c_l = 1 + sum_{t<l} s_t (s_t = the forward's saved dots) the recurrence unrolls to
//   x_l  = c_l x0 + sum_{t<l} b_t                      gx_l = g + sum_{t>l} ds_t w_t
//   ds_l = gx_l . x0 = g.x0 + sum_{t>l} ds_t (w_t.x0)  (scalars: L+1 dot products per sample, then a scalar recurrence)
//   dx   = c_L g + sum_l (ds_l c_l) w_l                (elementwise)
//   dw_l = sum_n (ds_l c_l)[n] x0[n]  +  (sum_{t<l} b_t) (sum_n ds_l[n])
//   db_l = sum_n g[n]  +  sum_{t>l} w_t (sum_n ds_t[n])
// so a wave keeps L+1 row accumulators (A_l = sum alpha_l x0 and G = sum g) instead of 2L, never rebuilds x_l, and the
// parameter-only terms are added once, in the reduction.  ~150 registers per lane instead of ~300: two waves per SIMD.
template <int NPL, int VEC, int LL>
__global__ __launch_bounds__(kDcnBwdThreads) void dcn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ s, const float* __restrict__ g,
                                                                 float* __restrict__ dx, float* __restrict__ partial, int B, int D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < LL * D; i += kDcnBwdThreads) smem[i] = w[i];
  __syncthreads();
  const float* ws = smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: descriptors built from it stay in SGPRs, no waterfall loops)
  float acc[LL][NPL], gsum[NPL], dssum[LL];
#pragma unroll
  for (int l = 0; l < LL; ++l) {
    dssum[l] = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) acc[l][i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) gsum[i] = 0.f;

  const int nstep = gridDim.x * kDcnBwdWaves;
  int n = blockIdx.x * kDcnBwdWaves + wave;
  float x0[NPL], gv[NPL];
  const int rowb = D * 4;
  float sv[LL];
  if (n < B) {
    row_load_buf<NPL, VEC>(dcn_rsrc(x + (long)n * D, rowb), lane, x0);
    row_load_buf<NPL, VEC>(dcn_rsrc(g + (long)n * D, rowb), lane, gv);
#pragma unroll
    for (int l = 0; l < LL; ++l) sv[l] = s[(long)n * LL + l];
  }
  for (; n < B; n += nstep) {
    // the next sample's rows AND its saved dots are fetched while this one is processed.  (The dots used to be loaded here,
    // behind the row prefetch: loads complete in order, so the wait for them also waited for the 2*NPL/VEC row loads just
    // issued -- an `s_waitcnt vmcnt(0)` in the middle of every sample, i.e. no prefetch at all.)
    float xn[NPL], gn[NPL], svn[LL];
    const bool more = n + nstep < B;
    {   // (past the last sample the descriptors are empty: the loads return zeros and nothing is fetched)
      const long nn = more ? n + nstep : n;
      row_load_buf<NPL, VEC>(dcn_rsrc(x + nn * D, more ? rowb : 0), lane, xn);
      row_load_buf<NPL, VEC>(dcn_rsrc(g + nn * D, more ? rowb : 0), lane, gn);
#pragma unroll
      for (int l = 0; l < LL; ++l) svn[l] = s[nn * LL + l];
    }
    float dots[LL + 1];
    dots[LL] = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) dots[LL] = fmaf(gv[i], x0[i], dots[LL]);
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      float wl[NPL];
      param_load<NPL, VEC>(ws + l * D, D, lane, wl);
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) t = fmaf(wl[i], x0[i], t);
      dots[l] = t;
    }
#pragma unroll
    for (int l = 0; l <= LL; ++l) dots[l] = wave_total(dots[l]);
    float ds[LL], alpha[LL], c = 1.f;
#pragma unroll
    for (int l = LL - 1; l >= 0; --l) {
      float t = dots[LL];
#pragma unroll
      for (int u = l + 1; u < LL; ++u) t = fmaf(ds[u], dots[u], t);
      ds[l] = t;
      dssum[l] += t;
    }
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      alpha[l] = ds[l] * c;
      c += sv[l];
    }
    float dxv[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      dxv[i] = gv[i] * c;
      gsum[i] += gv[i];
    }
    // (the parameter rows are read from LDS a second time on purpose: keeping the L rows of the dot phase alive would
    // cost L*NPL registers and a wave per SIMD; the laundered pointer stops the compiler from merging the two reads)
    int off2 = 0;                      // (an offset, not the pointer: the reads stay LDS reads, not flat ones)
    asm volatile("" : "+v"(off2));
    const float* ws2 = ws + off2;
#pragma unroll
    for (int l = 0; l < LL; ++l) {
      float wl[NPL];
      param_load<NPL, VEC>(ws2 + l * D, D, lane, wl);
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        dxv[i] = fmaf(wl[i], alpha[l], dxv[i]);
        acc[l][i] = fmaf(x0[i], alpha[l], acc[l][i]);
      }
    }
    row_store_buf<NPL, VEC>(dcn_rsrc(dx + (long)n * D, rowb), lane, dxv);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      x0[i] = xn[i];
      gv[i] = gn[i];
    }
#pragma unroll
    for (int l = 0; l < LL; ++l) sv[l] = svn[l];
  }

  // block reduction of the per-wave accumulators, fixed wave order, through LDS (one [waves][D] slab at a time):
  // partial[block] = [A_0 .. A_{L-1} | G | sum ds_0 .. sum ds_{L-1} (padded to 8)]
  __syncthreads();  // everyone is done with the staged params
  float* red = smem;  // needs kDcnBwdWaves*(D+8) floats (host sizes the dynamic LDS for max(params, this))
  float* pout = partial + (long)blockIdx.x * ((LL + 1) * D + 8);
#pragma unroll
  for (int l = 0; l <= LL; ++l) {
    if (l < LL) row_store<NPL, VEC>(red + wave * D, D, lane, acc[l < LL ? l : 0]);
    else row_store<NPL, VEC>(red + wave * D, D, lane, gsum);
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += kDcnBwdThreads) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < kDcnBwdWaves; ++wv) t += red[wv * D + d];
      pout[l * D + d] = t;
    }
    __syncthreads();
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < LL; ++l) red[wave * 8 + l] = dssum[l];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0.f;
    if (threadIdx.x < LL)
      for (int wv = 0; wv < kDcnBwdWaves; ++wv) t += red[wv * 8 + threadIdx.x];
    pout[(LL + 1) * D + threadIdx.x] = t;
  }
}

// dw, db from the workgroup partials of dcn_bwd_kernel (fixed order).  grid = (ceil(D/64), L+1): block (., l < L) sums A_l
// and the ds totals and writes dw_l = A_l + (sum_{t<l} b_t) DS_l; block (., L) sums G and writes every
// db_l = G + sum_{t>l} w_t DS_t.  The 4 waves take every 4th partial, wave sums are added in wave order.
__global__ __launch_bounds__(256) void dcn_reduce_closed_kernel(const float* __restrict__ partial, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ dw,
                                                                float* __restrict__ db, int parts, int D, int L) {
  __shared__ float red[4][64];
  __shared__ float dsr[4][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: descriptors built from it stay in SGPRs, no waterfall loops)
  const int d = blockIdx.x * 64 + lane, l = blockIdx.y;
  const long pstride = (long)(L + 1) * D + 8;
  // eight partials in flight per thread (they come from L2 / the Infinity Cache: the loop is latency-, not bandwidth-bound)
  float tq[8], dq[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) tq[u] = dq[u] = 0.f;
  // (branch-free: a partial index past the end is clamped and its value multiplied by 0, a column past D reads column D-1,
  // so the eight loads of a trip issue back to back)
  const int dc = min(d, D - 1), lc = min(lane, 7);
  for (int p0 = wave; p0 < parts; p0 += 32) {
    float a[8], e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* pp = partial + (long)min(p0 + 4 * u, parts - 1) * pstride;
      a[u] = pp[(long)l * D + dc];
      e[u] = pp[(long)(L + 1) * D + lc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float keep = p0 + 4 * u < parts ? 1.f : 0.f;
      tq[u] = fmaf(keep, a[u], tq[u]);
      dq[u] = fmaf(keep, e[u], dq[u]);
    }
  }
  const float t0 = ((tq[0] + tq[1]) + (tq[2] + tq[3])), t1 = ((tq[4] + tq[5]) + (tq[6] + tq[7]));
  const float dsl = ((dq[0] + dq[1]) + (dq[2] + dq[3])) + ((dq[4] + dq[5]) + (dq[6] + dq[7]));
  red[wave][lane] = t0 + t1;
  if (lane < 8) dsr[wave][lane] = dsl;
  __syncthreads();
  if (wave == 0 && d < D) {
    const float tot = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    auto DS = [&](int u) { return ((dsr[0][u] + dsr[1][u]) + dsr[2][u]) + dsr[3][u]; };
    if (l < L) {
      float bsum = 0.f;
      for (int u = 0; u < l; ++u) bsum += b[(long)u * D + d];
      dw[(long)l * D + d] = fmaf(bsum, DS(l), tot);
    } else {
      for (int ll = 0; ll < L; ++ll) {
        float v = tot;
        for (int u = ll + 1; u < L; ++u) v = fmaf(w[(long)u * D + d], DS(u), v);
        db[(long)ll * D + d] = v;
      }
    }
  }
}

// ---- generic backward for shapes whose per-lane accumulators do not fit the register file ---------------
// The backward recurrence only needs per-sample SCALARS besides elementwise work:
//   gx_l = g + sum_{t>l} w_t ds_t   =>   ds_l = g.x0 + sum_{t>l} ds_t (w_t.x0)
// pass 1 (wave per sample): a0 = g.x0, c_t = w_t.x0, then the scalar recurrence -> ds[n,l]
// pass 2 (thread per column d, loop over a chunk of samples): everything is elementwise given s[n,:], ds[n,:];
//         dw/db accumulate in 2*L registers per column; chunk partials are reduced by dcn_reduce_kernel.
constexpr int kDcnMaxL = 6;      // layers of the register-resident kernels
constexpr int kDcnGenMaxL = 16;  // layers of the generic (two-pass) kernels: any D

template <int LM>
__global__ __launch_bounds__(256) void dcn_bwd_scalars_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ g, float* __restrict__ ds, int B, int D,
                                                              int L) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: descriptors built from it stay in SGPRs, no waterfall loops)
  for (int n = blockIdx.x * 4 + wave; n < B; n += gridDim.x * 4) {
    float acc[LM + 1];
#pragma unroll
    for (int i = 0; i <= LM; ++i) acc[i] = 0.f;
    const float* xr = x + (long)n * D;
    const float* gr = g + (long)n * D;
    for (int d = lane; d < D; d += 64) {
      const float xv = xr[d];
      acc[0] = fmaf(gr[d], xv, acc[0]);
#pragma unroll
      for (int t = 0; t < LM; ++t)
        if (t < L) acc[t + 1] = fmaf(w[(long)t * D + d], xv, acc[t + 1]);
    }
#pragma unroll
    for (int i = 0; i <= LM; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
      float dsv[LM];
#pragma unroll
      for (int l = LM - 1; l >= 0; --l) {
        if (l < L) {
          float t = acc[0];
#pragma unroll
          for (int u = l + 1; u < LM; ++u)
            if (u < L) t = fmaf(dsv[u], acc[u + 1], t);
          dsv[l] = t;
          ds[(long)n * L + l] = t;
        } else {
          dsv[l] = 0.f;
        }
      }
    }
  }
}

template <int LM>
__global__ __launch_bounds__(256) void dcn_bwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, const float* __restrict__ s,
                                                           const float* __restrict__ ds, const float* __restrict__ g,
                                                           float* __restrict__ dx, float* __restrict__ partial, int B, int D, int L,
                                                           int nchunk) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_lo = blockIdx.y * nchunk, n_hi = min(B, n_lo + nchunk);
  float wl[LM], bl[LM], dwacc[LM], dbacc[LM];
#pragma unroll
  for (int l = 0; l < LM; ++l) {
    wl[l] = (d < D && l < L) ? w[(long)l * D + d] : 0.f;
    bl[l] = (d < D && l < L) ? b[(long)l * D + d] : 0.f;
    dwacc[l] = dbacc[l] = 0.f;
  }
  if (d < D) {
    for (int n = n_lo; n < n_hi; ++n) {
      const float x0 = x[(long)n * D + d];
      float gx = g[(long)n * D + d];
      float xl[LM];
      float cur = x0;
#pragma unroll
      for (int l = 0; l < LM; ++l) {
        xl[l] = cur;
        if (l < L) cur = fmaf(x0, s[(long)n * L + l], cur) + bl[l];
      }
      float dx0 = 0.f;
#pragma unroll
      for (int l = LM - 1; l >= 0; --l) {
        if (l < L) {
          const float dsl = ds[(long)n * L + l];
          dbacc[l] += gx;
          dx0 = fmaf(gx, s[(long)n * L + l], dx0);
          dwacc[l] = fmaf(xl[l], dsl, dwacc[l]);
          gx = fmaf(wl[l], dsl, gx);
        }
      }
      dx[(long)n * D + d] = dx0 + gx;
    }
    float* pout = partial + (long)blockIdx.y * 2 * L * D;
#pragma unroll
    for (int l = 0; l < LM; ++l) {
      if (l < L) {
        pout[(long)l * D + d] = dwacc[l];
        pout[(long)(L + l) * D + d] = dbacc[l];
      }
    }
  }
}


// ---- generic forward (any D, L <= kDcnGenMaxL): the closed form of the recurrence (reference interactive_layer.py:275-282).  With
// x_l = c_l x0 + sum_{t<l} b_t the layer's dot is s_l = c_l (x0 . w_l) + beta_l, beta_l = (sum_{t<l} b_t) . w_l (parameters only),
// c_{l+1} = c_l + s_l: ONE pass over the sample's row gives the L dots p_l = x0 . w_l (a wave per sample, L accumulators), the
// recurrence is L scalar steps, and the output is y = c_L x0 + sum_t b_t, column by column, in a second pass.  The register-resident
// kernel keeps a whole row in one wave (D <= 4096) and all parameters in LDS (2 L D 4 B <= 160 KiB); this pair has neither limit.
template <int LM>
__global__ __launch_bounds__(256) void dcn_fwd_scalars_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ s, int B, int D, int L) {
  __shared__ float beta[LM];
  __shared__ float red[4][LM];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {   // beta_l, once per workgroup (the grid is small: a persistent loop over the samples below)
    float acc[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l) acc[l] = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
      float bs = 0.f;
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < L) {
          acc[l] = fmaf(bs, w[(long)l * D + d], acc[l]);
          bs += b[(long)l * D + d];
        }
    }
#pragma unroll
    for (int l = 0; l < LM; ++l) {
      const float t = wave_sum(acc[l]);
      if (lane == 0) red[wave][l] = t;
    }
    __syncthreads();
    if (threadIdx.x < LM) beta[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
  for (int n = blockIdx.x * 4 + wave; n < B; n += gridDim.x * 4) {
    float p[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l) p[l] = 0.f;
    const float* xr = x + (long)n * D;
    for (int d = lane; d < D; d += 64) {
      const float xv = xr[d];
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < L) p[l] = fmaf(w[(long)l * D + d], xv, p[l]);
    }
#pragma unroll
    for (int l = 0; l < LM; ++l) p[l] = wave_sum(p[l]);
    if (lane == 0) {
      float c = 1.f;
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < L) {
          const float sl = fmaf(c, p[l], beta[l]);
          s[(long)n * L + l] = sl;
          c += sl;
        }
    }
  }
}
// y[n, d] = c_L[n] x0[n, d] + sum_t b_t[d],  c_L = 1 + sum_l s_l (in layer order, as the scalars kernel built it)
__global__ __launch_bounds__(256) void dcn_fwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ s,
                                                           float* __restrict__ y, int B, int D, int L, int nchunk) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_lo = blockIdx.y * nchunk, n_hi = min(B, n_lo + nchunk);
  if (d >= D) return;
  float bs = 0.f;
  for (int l = 0; l < L; ++l) bs += b[(long)l * D + d];
  for (int n = n_lo; n < n_hi; ++n) {
    float c = 1.f;
    for (int l = 0; l < L; ++l) c += s[(long)n * L + l];
    y[(long)n * D + d] = fmaf(c, x[(long)n * D + d], bs);
  }
}

// out[i] = sum over `parts` partials (fixed order).  One workgroup per 64 outputs: the 4 waves take every 4th partial
// (coalesced over i, four independent load streams), then the 4 wave sums are added in wave order.
__global__ __launch_bounds__(256) void dcn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                         float* __restrict__ db, int parts, int LD) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: descriptors built from it stay in SGPRs, no waterfall loops)
  const int i = blockIdx.x * 64 + lane;
  const int n = 2 * LD;
  float t0 = 0.f, t1 = 0.f;
  if (i < n) {
    int p = wave;
    for (; p + 4 < parts; p += 8) {
      t0 += partial[(long)p * n + i];
      t1 += partial[(long)(p + 4) * n + i];
    }
    if (p < parts) t0 += partial[(long)p * n + i];
  }
  red[wave][lane] = t0 + t1;
  __syncthreads();
  if (wave == 0 && i < n) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (i < LD) dw[i] = t;
    else db[i - LD] = t;
  }
}

// Workgroups: the forward likes many (1024: 20 us at c3), the backward few -- every workgroup writes one 2*L*D partial of
// dw/db that the reduction has to read back (256: bwd + reduce 49 us at c3, 512: 67 us, 1024: 105 us).
// FIL_DCN_GRID overrides both (tuning knob).
static int dcn_grid_cap(int B, int cap) {
  static const int forced = [] {   // read once, never on the launch path
    const char* e = getenv("FIL_DCN_GRID");
    return e != nullptr ? atoi(e) : 0;
  }();
  if (forced > 0) cap = forced;
  return cap;
}
static int dcn_grid(int B) { return std::max(1, std::min(cdiv(B, kDcnBwdWaves), dcn_grid_cap(B, 256))); }   // backward (and its workspace sizing)
static int dcn_grid_fwd(int B) { return std::max(1, std::min(cdiv(B, kDcnWaves), dcn_grid_cap(B, 1024))); }

static int pick_npl(int D, bool vec) {
  const int menu[] = {8, 20, 32, 64};
  for (int npl : menu)
    if ((long)npl * kWave >= D && (!vec || npl % 4 == 0)) return npl;
  return -1;
}

template <typename KernelT>
static void allow_lds(KernelT kernel, size_t sh) {
  if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
}

template <int NPL, int VEC>
static void launch_fwd(int grid, size_t sh, hipStream_t st, const float* x, const float* w, const float* b, float* y,
                       float* s, int B, int D, int L) {
  allow_lds(dcn_fwd_kernel<NPL, VEC>, sh);
  hipLaunchKernelGGL((dcn_fwd_kernel<NPL, VEC>), dim3(grid), dim3(kDcnThreads), sh, st, x, w, b, y, s, B, D, L);
}

template <int NPL, int VEC, int LL>
static void launch_bwd_l(int grid, size_t sh, hipStream_t st, const float* x, const float* w, const float* b,
                         const float* s, const float* g, float* dx, float* partial, int B, int D) {
  (void)b;
  allow_lds(dcn_bwd_kernel<NPL, VEC, LL>, sh);
  hipLaunchKernelGGL((dcn_bwd_kernel<NPL, VEC, LL>), dim3(grid), dim3(kDcnBwdThreads), sh, st, x, w, s, g, dx, partial, B, D);
}

template <int NPL, int VEC>
static int launch_bwd(int L, int grid, size_t sh, hipStream_t st, const float* x, const float* w, const float* b,
                      const float* s, const float* g, float* dx, float* partial, int B, int D) {
  switch (L) {
#define FIL_CASE(LL) case LL: launch_bwd_l<NPL, VEC, LL>(grid, sh, st, x, w, b, s, g, dx, partial, B, D); return 0;
    FIL_CASE(1) FIL_CASE(2) FIL_CASE(3) FIL_CASE(4) FIL_CASE(5) FIL_CASE(6)
#undef FIL_CASE
  }
  return -1;
}

constexpr size_t kDcnLdsLimit = 160 * 1024;  // whole LDS of one CU

}  // namespace fil

using namespace fil;

#define FIL_DCN_DISPATCH(NPLV, CALL_VEC, CALL_SCALAR)                \
  switch (NPLV) {                                                    \
    case 8: if (vec) { CALL_VEC(8); } else { CALL_SCALAR(8); } break;   \
    case 20: if (vec) { CALL_VEC(20); } else { CALL_SCALAR(20); } break; \
    case 32: if (vec) { CALL_VEC(32); } else { CALL_SCALAR(32); } break; \
    case 64: if (vec) { CALL_VEC(64); } else { CALL_SCALAR(64); } break; \
  }

extern "C" int fil_dcn_fwd(const float* x, const float* w, const float* b, float* y, float* s, int B, int D, int L,
                           void* stream) {
  FIL_CHECK_ARG(B >= 0 && D >= 1 && L >= 1);
  if (L > kDcnGenMaxL) return fail(FIL_ERR_UNSUPPORTED, "fil_dcn_fwd: L=%d > %d", L, kDcnGenMaxL);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(x && w && b && y && s);
  const bool vec = (D % 4 == 0);
  const int npl = pick_npl(D, vec);
  const size_t psz = (size_t)2 * L * D * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (L > kDcnMaxL || D > 4096 || npl <= 0 || psz > kDcnLdsLimit) {
    // outside the register-resident kernel's limits (a row in one wave, every parameter in LDS): the generic two-pass forward
    ProfScope ps("dcn_fwd_generic", st, (double)B * 3.0 * D * sizeof(float));
    const int grid = std::max(1, std::min(cdiv(B, 4), 2 * 256));
    if (L <= kDcnMaxL) hipLaunchKernelGGL(dcn_fwd_scalars_kernel<kDcnMaxL>, dim3(grid), dim3(256), 0, st, x, w, b, s, B, D, L);
    else hipLaunchKernelGGL(dcn_fwd_scalars_kernel<kDcnGenMaxL>, dim3(grid), dim3(256), 0, st, x, w, b, s, B, D, L);
    FIL_CHECK_LAUNCH();
    const int chunks = std::max(1, std::min(cdiv(B, 8), 256));
    const int nchunk = cdiv(B, chunks);
    hipLaunchKernelGGL(dcn_fwd_cols_kernel, dim3(cdiv(D, 256), cdiv(B, nchunk)), dim3(256), 0, st, x, b, s, y, B, D, L, nchunk);
    FIL_CHECK_LAUNCH();
    return FIL_OK;
  }
  const int grid = dcn_grid_fwd(B);
  ProfScope ps("dcn_fwd", st, (double)B * 2.0 * D * sizeof(float));
#define FWD_VEC(N) launch_fwd<N, 4>(grid, psz, st, x, w, b, y, s, B, D, L)
#define FWD_SCALAR(N) launch_fwd<N, 1>(grid, psz, st, x, w, b, y, s, B, D, L)
  FIL_DCN_DISPATCH(npl, FWD_VEC, FWD_SCALAR)
#undef FWD_VEC
#undef FWD_SCALAR
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

static bool dcn_register_path(int D, int L) {
  const int npl = pick_npl(D, D % 4 == 0);
  // per-lane accumulators are 2*L*NPL registers (+5*NPL of sample state): stay inside the 512-register file
  // closed-form backward: (L + 1) row accumulators + x0, g and their prefetched successors + a parameter row in flight,
  // two waves per SIMD (256 registers per lane)
  return L <= kDcnMaxL && D <= 4096 && npl > 0 && (L + 7) * npl <= 230 && std::max((size_t)L * D, (size_t)kDcnBwdWaves * (D + 8)) * sizeof(float) <= kDcnLdsLimit;
}
static int dcn_generic_chunks(int B) { return std::max(1, std::min(cdiv(B, 32), 64)); }

extern "C" size_t fil_dcn_bwd_workspace_bytes(int B, int D, int L) {
  if (B <= 0 || D <= 0 || L <= 0) return 0;
  if (dcn_register_path(D, L)) return align_up((size_t)dcn_grid(B) * ((size_t)(L + 1) * D + 8) * sizeof(float), 256);
  return align_up((size_t)B * L * sizeof(float), 256) + align_up((size_t)dcn_generic_chunks(B) * 2 * L * D * sizeof(float), 256);
}

extern "C" int fil_dcn_bwd(const float* x, const float* w, const float* b, const float* s, const float* g, float* dx,
                           float* dw, float* db, int B, int D, int L, void* workspace, size_t workspace_bytes,
                           void* stream) {
  FIL_CHECK_ARG(B >= 0 && D >= 1 && L >= 1);
  if (L > kDcnGenMaxL) return fail(FIL_ERR_UNSUPPORTED, "fil_dcn_bwd: L=%d > %d", L, kDcnGenMaxL);
  FIL_CHECK_ARG(dw && db);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)L * D * sizeof(float), st);
    (void)hipMemsetAsync(db, 0, (size_t)L * D * sizeof(float), st);
    return FIL_OK;
  }
  FIL_CHECK_ARG(x && w && b && s && g && dx);
  if (workspace == nullptr || workspace_bytes < fil_dcn_bwd_workspace_bytes(B, D, L))
    return fail(FIL_ERR_WORKSPACE, "fil_dcn_bwd: workspace %zu < %zu bytes", workspace_bytes,
                fil_dcn_bwd_workspace_bytes(B, D, L));
  if (!dcn_register_path(D, L)) {
    // generic two-pass backward (any D, L <= 16): scalars per sample, then column-wise accumulation
    Carver wsc(workspace);
    float* dsbuf = wsc.take<float>((size_t)B * L);
    float* partial = wsc.take<float>((size_t)dcn_generic_chunks(B) * 2 * L * D);
    ProfScope ps("dcn_bwd_generic", st, (double)B * 5.0 * D * sizeof(float));
    if (L <= kDcnMaxL) hipLaunchKernelGGL(dcn_bwd_scalars_kernel<kDcnMaxL>, dim3(std::min(cdiv(B, 4), 2048)), dim3(256), 0, st, x, w, g, dsbuf, B, D, L);
    else hipLaunchKernelGGL(dcn_bwd_scalars_kernel<kDcnGenMaxL>, dim3(std::min(cdiv(B, 4), 2048)), dim3(256), 0, st, x, w, g, dsbuf, B, D, L);
    FIL_CHECK_LAUNCH();
    const int chunks = dcn_generic_chunks(B);
    const int nchunk = cdiv(B, chunks);
    const dim3 grid(cdiv(D, 256), cdiv(B, nchunk));
    if (L <= kDcnMaxL) hipLaunchKernelGGL(dcn_bwd_cols_kernel<kDcnMaxL>, grid, dim3(256), 0, st, x, w, b, s, dsbuf, g, dx, partial, B, D, L, nchunk);
    else hipLaunchKernelGGL(dcn_bwd_cols_kernel<kDcnGenMaxL>, grid, dim3(256), 0, st, x, w, b, s, dsbuf, g, dx, partial, B, D, L, nchunk);
    FIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(dcn_reduce_kernel, dim3(cdiv(2 * L * D, 64)), dim3(256), 0, st, partial, dw, db, (int)grid.y, L * D);
    FIL_CHECK_LAUNCH();
    return FIL_OK;
  }
  const bool vec = (D % 4 == 0);
  const int npl = pick_npl(D, vec);
  const size_t psz = (size_t)L * D * sizeof(float);
  const size_t red = (size_t)kDcnBwdWaves * (D + 8) * sizeof(float);
  const size_t sh = std::max(psz, red);
  if (sh > kDcnLdsLimit) return fail(FIL_ERR_UNSUPPORTED, "fil_dcn_bwd: %zu bytes of LDS needed (> 160 KiB)", sh);
  const int grid = dcn_grid(B);
  float* partial = static_cast<float*>(workspace);
  int rc = 0;
  ProfScope ps("dcn_bwd", st, (double)B * 3.0 * D * sizeof(float));
#define BWD_VEC(N) rc = launch_bwd<N, 4>(L, grid, sh, st, x, w, b, s, g, dx, partial, B, D)
#define BWD_SCALAR(N) rc = launch_bwd<N, 1>(L, grid, sh, st, x, w, b, s, g, dx, partial, B, D)
  FIL_DCN_DISPATCH(npl, BWD_VEC, BWD_SCALAR)
#undef BWD_VEC
#undef BWD_SCALAR
  if (rc != 0) return fail(FIL_ERR_UNSUPPORTED, "fil_dcn_bwd: no kernel for L=%d", L);
  FIL_CHECK_LAUNCH();
  hipLaunchKernelGGL(dcn_reduce_closed_kernel, dim3(cdiv(D, 64), L + 1), dim3(256), 0, st, partial, w, b, dw, db, grid, D, L);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
