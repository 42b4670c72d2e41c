// What hides behind the f32-input MFMA on gfx950?  One wave per SIMD runs 8 v_mfma_f32_32x32x2_f32 per iteration plus N
// instructions of one other kind; prints the time per MFMA.  Result (round 1): nothing vector-side is free -- every v_fma
// costs its ~4 issue cycles of matrix time (the f32 MFMA runs on the vector ALUs).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value tools/probe_valu.hip -o /tmp/probe_valu && /tmp/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Kind { VALU = 0, VMEM = 1, LDS = 2, SALU = 3, VADDR = 4, PKFMA = 5, ACCREAD = 6 };
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float* out, const float* __restrict__ src, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float v[N > 0 ? N : 1];
  for (int i = 0; i < (N > 0 ? N : 1); ++i) v[i] = seed + i;
  const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
  f32x2 pk[N > 0 ? N : 1];
  for (int i = 0; i < (N > 0 ? N : 1); ++i) pk[i] = f32x2{seed + i, seed - i};
  const f32x2 pka = {a, b}, pkb = {b, a};
  int so = 0;
  unsigned long long vaddr = (unsigned long long)src + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int k = m * N / 8; k < (m + 1) * N / 8; ++k) {
        if constexpr (KIND == VALU) v[k] = __builtin_fmaf(v[k], b, a);
        if constexpr (KIND == VMEM) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (threadIdx.x & 63) * 4, so + k * 256, 0));
        if constexpr (KIND == LDS) v[k] = lds[(threadIdx.x + k * 64 + it) & 4095];
        if constexpr (KIND == SALU) { so = (so * 3 + k) & 0xffff; asm volatile("" : "+s"(so)); }
        if constexpr (KIND == VADDR) { vaddr += 1024; asm volatile("" : "+v"(vaddr)); }
        if constexpr (KIND == PKFMA) { pk[k] = __builtin_elementwise_fma(pk[k], pkb, pka); }
        if constexpr (KIND == ACCREAD) { v[k] += acc[(m + 2) & 3][k & 15]; }
      }
    }
    if constexpr (KIND == VMEM) so = (so + 4096) & 0xffff;
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = (float)so + (float)(vaddr & 0xff);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < (N > 0 ? N : 1); ++i) s += v[i] + pk[i][0] + pk[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int N>
static void run(float* d, const float* src, int iters, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<KIND, N><<<256, 256>>>(d, src, iters, 1.0f);
  hipEventRecord(e0);
  probe<KIND, N><<<256, 256>>>(d, src, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 8.0 * iters;
  printf("%-28s %2d per 8 MFMA   %.1f ns per MFMA   %.1f TFLOP/s\n", what, N, ms * 1e6 / mfmas, 256.0 * 4 * mfmas * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float *d, *src;
  hipMalloc(&d, 256 * 256 * 4);
  hipMalloc(&src, 1 << 21);
  hipMemset(src, 0, 1 << 21);
  const int iters = 20000;
  run<VALU, 0>(d, src, iters, "MFMA only");
  run<VALU, 8>(d, src, iters, "v_fma_f32");
  run<VALU, 16>(d, src, iters, "v_fma_f32");
  run<VMEM, 2>(d, src, iters, "buffer_load_dword (soffset)");
  run<VMEM, 8>(d, src, iters, "buffer_load_dword (soffset)");
  run<LDS, 2>(d, src, iters, "ds_read_b32");
  run<LDS, 8>(d, src, iters, "ds_read_b32");
  run<SALU, 8>(d, src, iters, "s_mul/s_add/s_and");
  run<SALU, 32>(d, src, iters, "s_mul/s_add/s_and");
  run<VADDR, 8>(d, src, iters, "64-bit v_add (address math)");
  run<PKFMA, 8>(d, src, iters, "v_pk_fma_f32");
  run<PKFMA, 16>(d, src, iters, "v_pk_fma_f32");
  run<ACCREAD, 8>(d, src, iters, "v_accvgpr_read + v_add");
  return 0;
}
