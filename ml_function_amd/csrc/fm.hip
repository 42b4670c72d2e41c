// A1  FM second-order interaction (+ N3 pair list) for gfx950.
//
// Replaces InnerLayer.call / FmLayer.call of the reference
// (kon/model/ctr_model/layer/interactive_layer/interactive_layer.py:59-66,161-170): C(F,2) tf.multiply
// ops + an N-ary Add + the Add of the linear terms.  Pure HBM streaming (6 flops per 4 bytes), so the kernels are built
// around the memory system:
//   * a wave owns a tile of whole samples; their [F,K] slabs are one contiguous byte range that goes HBM -> LDS by
//     LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB contiguous per wave instruction, no VGPRs, out-of-range lanes write
//     zeros), all of a tile's pieces in flight before the first is waited for;
//   * compute runs out of LDS: lane (sample, 4 consecutive k) walks the fields with the prefix form
//     acc += e_f * run; run += e_f (the same F(F-1)/2 products as the reference, no (sum^2 - sum of squares)
//     cancellation), fp32 accumulation for fp32 and bf16 storage;
//   * outputs leave as whole rows: out [B,K] is contiguous over a wave's lanes, demb is written field-row by field-row
//     (64/(K/4) fields = 1 KiB per store instruction at K = 16).
// Round 1's thread-per-(sample, k4) kernels read 16 different 64-byte segments at a 2,496-byte stride per wave instruction
// (0.47-0.49 of the 8 TB/s peak at B = 1 M); they remain as the K % 4 != 0 fallback.
#include "common.h"
#include <algorithm>

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

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged streaming kernels (K % 4 == 0).  4 waves per workgroup, each with its own LDS region and its own tiles.
constexpr int kFmWaves = 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fm_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL), 0x00020000);
}

// bytes [0, nbytes) of the buffer -> LDS at dst (linear image), 1 KiB per wave instruction; pieces beyond nbytes come back 0
__device__ __forceinline__ void fm_dma_tile(__amdgpu_buffer_rsrc_t r, unsigned char* dst, int nbytes, int lane) {
  for (int off = 0; off < nbytes; off += 1024)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + off), 16, off + lane * 16, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void fm_lds_load4(const T* p, float (&v)[4]) { VecIO<T, 4>::load(p, v); }

// sizes of a wave's LDS regions (bytes, 16-byte aligned): embedding tile | linear terms or g | S | per-(sample, k4) partials
struct FmTile {
  int TS;        // samples per tile
  int e_bytes, l_bytes, s_bytes, p_bytes;
  __host__ __device__ int wave_bytes() const { return e_bytes + l_bytes + s_bytes + p_bytes; }
};

template <typename T>
__global__ __launch_bounds__(64 * kFmWaves) void fm_fwd_lds_kernel(const T* __restrict__ emb, const float* __restrict__ lin,
                                                                  T* __restrict__ out, int B, int F, int K, FmTile tl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* base = fm_smem + wave * tl.wave_bytes();
  const T* tile = reinterpret_cast<const T*>(base);
  const float* ltile = reinterpret_cast<const float*>(base + tl.e_bytes);
  const int FK = F * K, KQ = K >> 2;
  const int ntiles = (B + tl.TS - 1) / tl.TS;
  const int kq = lane % KQ, bl0 = lane / KQ, bstep = 64 / KQ;      // lane -> (sample within the tile, 4 consecutive k)
  for (int t = blockIdx.x * kFmWaves + wave; t < ntiles; t += gridDim.x * kFmWaves) {
    const int b0 = t * tl.TS, nb = min(tl.TS, B - b0);
    // (descriptors run to the end of the tensor: the last 16-byte piece of a tile may reach into the next sample)
    fm_dma_tile(fm_rsrc(emb + (long)b0 * FK, (long)(B - b0) * FK * sizeof(T)), base, nb * FK * (int)sizeof(T), lane);
    if (lin != nullptr) fm_dma_tile(fm_rsrc(lin + (long)b0 * F, (long)(B - b0) * F * 4), base + tl.e_bytes, nb * F * 4, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int bl = bl0; bl < nb && bl0 < bstep; bl += bstep) {
      const T* p = tile + bl * FK + kq * 4;
      float run[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int f = 0; f < F; ++f) {
        float e[4];
        fm_lds_load4(p + f * K, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = fmaf(e[i], run[i], acc[i]);
          run[i] += e[i];
        }
      }
      float ls = 0.f;
      if (lin != nullptr) {
        const float* lp = ltile + bl * F;
        for (int f = 0; f < F; ++f) ls += lp[f];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += ls;
      VecIO<T, 4>::store(out + (long)(b0 + bl) * K + kq * 4, acc);
    }
    // the next tile's DMA overwrites this image: every LDS read above has returned (their values were stored)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <typename T>
__global__ __launch_bounds__(64 * kFmWaves) void fm_bwd_lds_kernel(const T* __restrict__ emb, const T* __restrict__ g,
                                                                  T* __restrict__ demb, float* __restrict__ dlin, int B, int F,
                                                                  int K, FmTile tl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* base = fm_smem + wave * tl.wave_bytes();
  const T* tile = reinterpret_cast<const T*>(base);
  const T* gtile = reinterpret_cast<const T*>(base + tl.e_bytes);
  float* stile = reinterpret_cast<float*>(base + tl.e_bytes + tl.l_bytes);              // S[b][k], fp32
  float* ptile = reinterpret_cast<float*>(base + tl.e_bytes + tl.l_bytes + tl.s_bytes);  // sum of 4 g per (b, k4)
  const int FK = F * K, KQ = K >> 2;
  const int ntiles = (B + tl.TS - 1) / tl.TS;
  const int kq = lane % KQ, bl0 = lane / KQ, bstep = 64 / KQ;
  const int fpi = 64 / KQ;    // field rows per store instruction
  for (int t = blockIdx.x * kFmWaves + wave; t < ntiles; t += gridDim.x * kFmWaves) {
    const int b0 = t * tl.TS, nb = min(tl.TS, B - b0);
    fm_dma_tile(fm_rsrc(emb + (long)b0 * FK, (long)(B - b0) * FK * sizeof(T)), base, nb * FK * (int)sizeof(T), lane);
    fm_dma_tile(fm_rsrc(g + (long)b0 * K, (long)(B - b0) * K * sizeof(T)), base + tl.e_bytes, nb * K * (int)sizeof(T), lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // pass 1: S[b,k] = sum_f e[b,f,k]; partial sums of g for dlin
    for (int bl = bl0; bl < nb && bl0 < bstep; bl += bstep) {
      const T* p = tile + bl * FK + kq * 4;
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int f = 0; f < F; ++f) {
        float e[4];
        fm_lds_load4(p + f * K, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += e[i];
      }
      *reinterpret_cast<float4*>(stile + bl * K + kq * 4) = make_float4(s[0], s[1], s[2], s[3]);
      float gv[4];
      fm_lds_load4(gtile + bl * K + kq * 4, gv);
      ptile[bl * KQ + kq] = (gv[0] + gv[1]) + (gv[2] + gv[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // pass 2: demb[b,f,:] = g (S - e), field rows in order (contiguous stores); lane -> (field within the group, k4)
    for (int bl = 0; bl < nb; ++bl) {
      float gv[4];
      fm_lds_load4(gtile + bl * K + kq * 4, gv);
      const float4 sv = *reinterpret_cast<const float4*>(stile + bl * K + kq * 4);
      const float sa[4] = {sv.x, sv.y, sv.z, sv.w};
      T* q = demb + (long)(b0 + bl) * FK + kq * 4;
      for (int f = bl0; f < F && bl0 < fpi; f += fpi) {
        float e[4], dd[4];
        fm_lds_load4(tile + bl * FK + f * K + kq * 4, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) dd[i] = gv[i] * (sa[i] - e[i]);
        VecIO<T, 4>::store(q + f * K, dd);
      }
      if (dlin != nullptr) {    // dlin[b,f] = sum_k g[b,k] for every f
        float gs = 0.f;
        for (int i = 0; i < KQ; ++i) gs += ptile[bl * KQ + i];
        for (int f = lane; f < F; f += 64) dlin[(long)(b0 + bl) * F + f] = gs;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// tile shape: ~12 KiB of embeddings per wave (eight waves of tiles in flight per CU cover the HBM latency).  Measured at
// B = 1 M, F = 39, K = 16 fp32 (profiles/r02_fm_tile_sweep.txt): this shape 5.44 / 5.29 TB/s fwd / bwd; two buffers per wave
// with the next tile's DMA under the current compute 3.7-4.1 / 5.2 (half the waves per CU); splitting a sample's field walk
// over lane groups to fill all 64 lanes 5.1 / 5.2 (the extra LDS round trip costs more than the idle lanes).
static FmTile fm_tile(int F, int K, int esz, bool bwd) {
  FmTile tl;
  const int slab = F * K * esz;
  tl.TS = std::max(1, std::min(64, 12288 / std::max(slab, 1)));
  auto al = [](int v) { return (v + 15) / 16 * 16; };
  tl.e_bytes = al(tl.TS * slab) + 1024;                      // + slack: the last 1-KiB piece of the DMA writes whole
  tl.l_bytes = al(tl.TS * (bwd ? K * esz : F * 4)) + 1024;
  tl.s_bytes = bwd ? al(tl.TS * K * 4) : 0;
  tl.p_bytes = bwd ? al(tl.TS * (K / 4) * 4) : 0;
  return tl;
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
  // (K * sizeof(T) % 16 == 0: every sample slab, and so every tile and the tensor itself, is a whole number of 16-byte DMA pieces)
  if (K % 4 == 0 && (K * sizeof(T)) % 16 == 0 && K / 4 <= 64 && (long)F * K * sizeof(T) <= 32768) {
    const FmTile tl = fm_tile(F, K, (int)sizeof(T), !fwd);
    const size_t sh = (size_t)kFmWaves * tl.wave_bytes();
    const int ntiles = (B + tl.TS - 1) / tl.TS;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / sh));
    const int grid = std::max(1, std::min((ntiles + kFmWaves - 1) / kFmWaves, 256 * per_cu));
    if (fwd) {
      if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fm_fwd_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipLaunchKernelGGL((fm_fwd_lds_kernel<T>), dim3(grid), dim3(64 * kFmWaves), sh, st, (const T*)emb, (const float*)lin_or_g, (T*)out, B, F, K, tl);
    } else {
      if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fm_bwd_lds_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipLaunchKernelGGL((fm_bwd_lds_kernel<T>), dim3(grid), dim3(64 * kFmWaves), sh, st, (const T*)emb, (const T*)lin_or_g, (T*)out, dlin, B, F, K, tl);
    }
    return 0;
  }
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
