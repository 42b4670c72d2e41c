// Feasibility probe for a split-bf16 ("bf16x3") CIN: per iteration 24 v_mfma_f32_32x32x16_bf16 (4 accumulators x 6 split
// products = the work that replaces 32 f32 MFMAs) plus NV v_fma_f32 standing in for the operand-split VALU work.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value tools/probe_bf16x3.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int CHAIN = 1>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float v[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = seed + i;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x + i); b[i] = (__bf16)(seed * 0.5f + i); }
  const float fa = seed * 0.5f, fb = seed * 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      acc[(m / CHAIN) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(m / CHAIN) & 3], 0, 0, 0);  // CHAIN consecutive MFMAs share an accumulator
#pragma unroll
      for (int k = m * NV / 24; k < (m + 1) * NV / 24; ++k) v[k] = __builtin_fmaf(v[k], fb, fa);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int CHAIN = 1>
static void run(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<NV, CHAIN><<<256, 256>>>(d, iters, 1.0f);
  hipEventRecord(e0);
  probe<NV, CHAIN><<<256, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // one iteration replaces 32 f32 MFMAs (4096 flop each per SIMD-wave) of the exact-fp32 kernels
  const double equiv = 256.0 * 4 * 32.0 * iters * 4096.0 / (ms * 1e-3) / 1e12;
  printf("chain %d  VALU per 24 bf16 MFMA: %3d   %.1f ns per group   fp32-equivalent %.0f TFLOP/s   (bf16 executed %.0f TFLOP/s)\n", CHAIN, NV,
         ms * 1e6 / iters, equiv, 256.0 * 4 * 24.0 * iters * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  const int iters = 20000;
  run<0>(d, iters);
  run<24>(d, iters);
  run<48>(d, iters);
  run<72>(d, iters);
  run<96>(d, iters);
  run<144>(d, iters);
  run<0, 6>(d, iters);
  run<48, 6>(d, iters);
  run<0, 24>(d, iters);
  return 0;
}
