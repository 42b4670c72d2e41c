// N1  SparseEmbed field-index work for gfx950: packed [B,F,K] gather and its scatter-add gradient.
//
// Replaces the F separate Embedding lookups + Concatenate of the reference
// (interactive_layer.py:225-242, models.py:131): one launch emits the packed layout the interaction
// kernels consume.  Rows are copied bit-exactly.  A (b,f) row of K floats is moved by K/4 lanes with
// 16-byte accesses; consecutive lanes walk consecutive rows of the output, so stores are fully coalesced
// and each gathered table row is read as one contiguous 4*K-byte segment.
#include "common.h"

namespace fil {

template <int VEC>
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ offsets,
                                                           const int64_t* __restrict__ idx, float* __restrict__ out, long rows,
                                                           int F, int K) {
  const int KV = K / VEC;
  const long total = rows * KV;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long row = t / KV;
    const int kv = (int)(t - row * KV);
    const int f = (int)(row % F);
    const long src = (offsets[f] + idx[row]) * K + (long)kv * VEC;
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(out + row * K + kv * 4) = *reinterpret_cast<const float4*>(table + src);
    } else {
      out[row * K + kv] = table[src];
    }
  }
}

// dtable[offsets[f] + idx[b,f], k] += g[b,f,k]  -- fp32 global atomics (one dword per lane, contiguous per row).
// The order of additions into a row that is hit several times in a batch is not fixed.
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ idx,
                                                            const float* __restrict__ g, float* __restrict__ dtable, long rows,
                                                            int F, int K) {
  const long total = rows * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long row = t / K;
    const int k = (int)(t - row * K);
    const int f = (int)(row % F);
    atomicAdd(dtable + (offsets[f] + idx[row]) * K + k, g[t]);
  }
}

}  // namespace fil

using namespace fil;

extern "C" int fil_embed_gather(const float* table, const int64_t* offsets, const int64_t* idx, float* out, int B, int F, int K,
                                void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(table && offsets && idx && out);
  const long rows = (long)B * F;
  const bool vec = (K % 4 == 0);
  const long total = rows * (vec ? K / 4 : K);
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 8);
  if (vec) hipLaunchKernelGGL((embed_gather_kernel<4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table, offsets, idx, out, rows, F, K);
  else hipLaunchKernelGGL((embed_gather_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table, offsets, idx, out, rows, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_scatter_add(const int64_t* offsets, const int64_t* idx, const float* g, float* dtable, int B, int F,
                                     int K, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(offsets && idx && g && dtable);
  const long rows = (long)B * F;
  const long total = rows * K;
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, offsets, idx, g, dtable, rows, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
