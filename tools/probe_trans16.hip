// Are the f16 transcendentals (v_exp_f16 / v_rcp_f16) issued faster than the f32 ones on gfx950?  8 per iteration, 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_trans16.hip -o tools/bin/probe_trans16 && tools/bin/probe_trans16
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  float t[8];
  _Float16 h[8];
  for (int i = 0; i < 8; ++i) { t[i] = seed + 0.01f * i + threadIdx.x * 1e-4f; h[i] = (_Float16)t[i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (KIND == 0) { if (k & 1) t[k] = __builtin_amdgcn_rcpf(t[k]); else t[k] = __builtin_amdgcn_exp2f(t[k]); }
      if (KIND == 1) { if (k & 1) asm volatile("v_rcp_f16 %0, %0" : "+v"(h[k])); else asm volatile("v_exp_f16 %0, %0" : "+v"(h[k])); }
      if (KIND == 2) { if (k & 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(t[k])); else asm volatile("v_exp_f32 %0, %0" : "+v"(t[k])); }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += t[i] + (float)h[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
static void run(float* d, const char* what) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  probe<KIND><<<256, 512, 96 * 1024>>>(d, 100, 1.0f);
  hipEventRecord(e0);
  probe<KIND><<<256, 512, 96 * 1024>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7.1f cycles per 8 per SIMD (2 waves)  (%.3f ms)\n", what, ms * 1e-3 * 2.4e9 / iters / 2, ms);
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 512 * sizeof(float));
  run<0>(d, "f32 exp2/rcp (builtins)");
  run<2>(d, "f32 exp/rcp (asm)");
  run<1>(d, "f16 exp/rcp (asm)");
  return 0;
}
