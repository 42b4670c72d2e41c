// A1  FM second-order interaction (+ N3 pair list) for gfx950.
//
// Replaces InnerLayer.call / FmLayer.call of the reference
// (kon/model/ctr_model/layer/interactive_layer/interactive_layer.py:59-66,161-170): C(F,2) tf.multiply
// ops + an N-ary Add + the Add of the linear terms.  HBM-streaming: a thread owns VEC consecutive k of
// one sample and walks the sample's contiguous [F,K] slab with VEC-wide loads (16 B for fp32, 8 B for
// bf16); the K/VEC threads of a sample read whole 64..128-byte rows, so every fetched line is consumed.
// Forward uses the prefix form acc += e_f * run; run += e_f (the same F(F-1)/2 products as the
// reference, no (sum^2 - sum of squares) cancellation).  Accumulation is fp32 for every storage dtype.
#include "common.h"

namespace fil {

template <typename T, int VEC>
struct VecIO;

template <int VEC>
struct VecIO<float, VEC> {
  __device__ static void load(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = p[i];
    }
  }
  __device__ static void store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) p[i] = v[i];
    }
  }
};

// bf16 is handled as raw 16-bit storage; conversion to/from fp32 by the hardware cast.
template <int VEC>
struct VecIO<__hip_bfloat16, VEC> {
  __device__ static void load(const __hip_bfloat16* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
      const uint2 t = *reinterpret_cast<const uint2*>(p);
      v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
      v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = __bfloat162float(p[i]);
    }
  }
  __device__ static void store(__hip_bfloat16* p, const float (&v)[VEC]) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = __float2bfloat16(v[i]);
  }
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void fm_fwd_kernel(const T* __restrict__ emb, const float* __restrict__ lin,
                                                     T* __restrict__ out, int B, int F, int K) {
  const int KV = K / VEC;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)B * KV) return;
  const int b = (int)(tid / KV), kv = (int)(tid % KV);
  const T* p = emb + (long)b * F * K + kv * VEC;
  float run[VEC], acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) run[i] = acc[i] = 0.f;
#pragma unroll 4
  for (int f = 0; f < F; ++f) {
    float e[VEC];
    VecIO<T, VEC>::load(p + (long)f * K, e);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      acc[i] = fmaf(e[i], run[i], acc[i]);
      run[i] += e[i];
    }
  }
  float ls = 0.f;
  if (lin != nullptr) {
    const float* lp = lin + (long)b * F;
    for (int f = 0; f < F; ++f) ls += lp[f];
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] += ls;
  VecIO<T, VEC>::store(out + (long)b * K + kv * VEC, acc);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void fm_bwd_kernel(const T* __restrict__ emb, const T* __restrict__ g,
                                                     T* __restrict__ demb, float* __restrict__ dlin, int B, int F,
                                                     int K) {
  const int KV = K / VEC;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long)B * KV) return;
  const int b = (int)(tid / KV), kv = (int)(tid % KV);
  const T* p = emb + (long)b * F * K + kv * VEC;
  T* q = demb + (long)b * F * K + kv * VEC;
  float s[VEC], gv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
#pragma unroll 4
  for (int f = 0; f < F; ++f) {
    float e[VEC];
    VecIO<T, VEC>::load(p + (long)f * K, e);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] += e[i];
  }
  VecIO<T, VEC>::load(g + (long)b * K + kv * VEC, gv);
#pragma unroll 4
  for (int f = 0; f < F; ++f) {  // second pass hits L1/L2 (the slab was just read by this thread)
    float e[VEC], d[VEC];
    VecIO<T, VEC>::load(p + (long)f * K, e);
#pragma unroll
    for (int i = 0; i < VEC; ++i) d[i] = gv[i] * (s[i] - e[i]);
    VecIO<T, VEC>::store(q + (long)f * K, d);
  }
  if (dlin != nullptr) {  // dlin[b,f] = sum_k g[b,k] for every f; the sample's KV threads split the f's
    float gs = 0.f;
    const T* gp = g + (long)b * K;
    for (int k = 0; k < K; k += VEC) {
      float t[VEC];
      VecIO<T, VEC>::load(gp + k, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) gs += t[i];
    }
    for (int f = kv; f < F; f += KV) dlin[(long)b * F + f] = gs;
  }
}

// pair index p (combinations order: i ascending, then j ascending) -> (i, j)
__device__ __forceinline__ void pair_from_index(int p, int F, int& i, int& j) {
  const float t = (float)(2 * F - 1);
  int ii = (int)((t - sqrtf(t * t - 8.f * (float)p)) * 0.5f);
  ii = max(0, min(ii, F - 2));
  while (ii > 0 && ii * (2 * F - ii - 1) / 2 > p) --ii;
  while ((ii + 1) * (2 * F - ii - 2) / 2 <= p) ++ii;
  i = ii;
  j = p - ii * (2 * F - ii - 1) / 2 + ii + 1;
}

__device__ __forceinline__ int pair_index(int i, int j, int F) { return i * (2 * F - i - 1) / 2 + (j - i - 1); }

__global__ __launch_bounds__(256) void fm_pairs_fwd_kernel(const float* __restrict__ emb, float* __restrict__ pairs,
                                                           int B, int F, int K) {
  const int P = F * (F - 1) / 2;
  const long total = (long)B * P * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int k = (int)(t % K);
    const long bp = t / K;
    const int p = (int)(bp % P);
    const int b = (int)(bp / P);
    int i, j;
    pair_from_index(p, F, i, j);
    const float* e = emb + (long)b * F * K;
    pairs[t] = e[i * K + k] * e[j * K + k];
  }
}

__global__ __launch_bounds__(256) void fm_pairs_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ gp,
                                                           float* __restrict__ demb, int B, int F, int K) {
  const int P = F * (F - 1) / 2;
  const long total = (long)B * F * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int k = (int)(t % K);
    const long bf = t / K;
    const int f = (int)(bf % F);
    const int b = (int)(bf / F);
    const float* e = emb + (long)b * F * K;
    const float* gq = gp + (long)b * P * K;
    float acc = 0.f;
    for (int j = 0; j < f; ++j) acc = fmaf(gq[(long)pair_index(j, f, F) * K + k], e[j * K + k], acc);
    for (int j = f + 1; j < F; ++j) acc = fmaf(gq[(long)pair_index(f, j, F) * K + k], e[j * K + k], acc);
    demb[t] = acc;
  }
}

template <typename T>
static int launch_fm(bool fwd, const void* emb, const void* lin_or_g, void* out, float* dlin, int B, int F, int K,
                     hipStream_t st) {
  const bool vec = (K % 4 == 0);
  const long threads = (long)B * (vec ? K / 4 : K);
  const int grid = (int)((threads + 255) / 256);
  if (fwd) {
    if (vec)
      hipLaunchKernelGGL((fm_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)emb, (const float*)lin_or_g,
                         (T*)out, B, F, K);
    else
      hipLaunchKernelGGL((fm_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)emb, (const float*)lin_or_g,
                         (T*)out, B, F, K);
  } else {
    if (vec)
      hipLaunchKernelGGL((fm_bwd_kernel<T, 4>), dim3(grid), dim3(256), 0, st, (const T*)emb, (const T*)lin_or_g,
                         (T*)out, dlin, B, F, K);
    else
      hipLaunchKernelGGL((fm_bwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, (const T*)emb, (const T*)lin_or_g,
                         (T*)out, dlin, B, F, K);
  }
  return 0;
}

}  // namespace fil

using namespace fil;

extern "C" int fil_fm_fwd(const void* emb, const float* lin, void* out, int B, int F, int K, int dtype, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  FIL_CHECK_ARG(dtype == FIL_F32 || dtype == FIL_BF16);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(emb != nullptr && out != nullptr);
  hipStream_t st = (hipStream_t)stream;
  const double esz = dtype == FIL_F32 ? 4.0 : 2.0;
  ProfScope ps("fm_fwd", st, (double)B * ((double)F * K * esz + K * esz + (lin ? F * 4.0 : 0.0)));
  if (dtype == FIL_F32) launch_fm<float>(true, emb, lin, out, nullptr, B, F, K, st);
  else launch_fm<__hip_bfloat16>(true, emb, lin, out, nullptr, B, F, K, st);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_fm_bwd(const void* emb, const void* g, void* demb, float* dlin, int B, int F, int K, int dtype,
                          void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  FIL_CHECK_ARG(dtype == FIL_F32 || dtype == FIL_BF16);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(emb != nullptr && g != nullptr && demb != nullptr);
  hipStream_t st = (hipStream_t)stream;
  const double esz = dtype == FIL_F32 ? 4.0 : 2.0;
  ProfScope ps("fm_bwd", st, (double)B * (2.0 * F * K * esz + K * esz + (dlin ? F * 4.0 : 0.0)));
  if (dtype == FIL_F32) launch_fm<float>(false, emb, g, demb, dlin, B, F, K, st);
  else launch_fm<__hip_bfloat16>(false, emb, g, demb, dlin, B, F, K, st);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_fm_pairs_fwd(const float* emb, float* pairs, int B, int F, int K, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 2 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(emb != nullptr && pairs != nullptr);
  const long total = (long)B * (F * (F - 1) / 2) * K;
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(fm_pairs_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, emb, pairs, B, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_fm_pairs_bwd(const float* emb, const float* gpairs, float* demb, int B, int F, int K, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 2 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(emb != nullptr && gpairs != nullptr && demb != nullptr);
  const long total = (long)B * F * K;
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(fm_pairs_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, emb, gpairs, demb, B, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
