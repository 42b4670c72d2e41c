// How many score tiles per cycle can one SIMD of gfx950 sustain on the instruction MIX of the attention kernels' tile loops, at
// 1, 2, 3, 4 waves per SIMD?  No memory, no LDS: per "tile" a wave issues NT transcendentals (v_exp_f32 / v_rcp_f32 alternating),
// NV plain VALU (v_fma_f32 on independent chains), NC v_cvt_pk f16 conversions and NM v_mfma_f32_16x16x16_f16.
// Prints cycles per tile per SIMD (= kernel cycles / tiles issued on that SIMD).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_tile.hip -o /tmp/probe_tile && /tmp/probe_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
// PK: the NV plain operations as NV / 2 packed v_pk_fma_f32 (two elements per instruction) instead of NV v_fma_f32
template <int NT, int NV, int NC, int NM, int WPS, bool PK = false>
__global__ __launch_bounds__(256 * WPS) void probe(float* out, int iters, float seed) {
  float t[8], v[8];
  f32x4 acc[4];
  f16x4 op = {(_Float16)seed, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f};
  for (int i = 0; i < 8; ++i) { t[i] = seed + 0.01f * i + threadIdx.x * 1e-4f; v[i] = seed * 0.5f + i; }
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 0.999f, b = 1e-3f;
  f16x2 cv[4] = {};
  f32x2 pv[4];
  for (int i = 0; i < 4; ++i) pv[i] = f32x2{seed + i, seed - i};
  const f32x2 pa = {a, a}, pb = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (k & 1) t[k & 7] = __builtin_amdgcn_rcpf(t[k & 7]); else t[k & 7] = __builtin_amdgcn_exp2f(t[k & 7]);
    }
    if constexpr (PK) {
#pragma unroll
      for (int k = 0; k < NV / 2; ++k) pv[k & 3] = __builtin_elementwise_fma(pv[k & 3], pa, pb);
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], a, b);
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) cv[k & 3] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[k & 7], t[k & 7]));
#pragma unroll
    for (int k = 0; k < NM; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(op, op, acc[k & 3], 0, 0, 0);
    if (NC > 0) { op[0] += cv[0][0]; }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += t[i] + v[i];
  for (int i = 0; i < 4; ++i) s += pv[i][0] + pv[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)cv[i][0] + (float)cv[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int NV, int NC, int NM, int WPS, bool PK = false>
static void run(float* d, const char* what) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // 96 KB of dynamic LDS per workgroup: exactly one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NT, NV, NC, NM, WPS, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  probe<NT, NV, NC, NM, WPS, PK><<<256, 256 * WPS, 96 * 1024>>>(d, 100, 1.0f);
  hipEventRecord(e0);
  probe<NT, NV, NC, NM, WPS, PK><<<256, 256 * WPS, 96 * 1024>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double cyc = ms * 1e-3 * clk_khz * 1e3;
  printf("%-34s waves/SIMD %d: %8.1f cycles per tile per wave, %8.1f per tile per SIMD  (%.3f ms, clock %d MHz nominal)\n", what, WPS,
         cyc / iters, cyc / iters / WPS, ms, clk_khz / 1000);
}

#define ALLW(NT, NV, NC, NM, WHAT) \
  run<NT, NV, NC, NM, 1>(d, WHAT); run<NT, NV, NC, NM, 2>(d, WHAT); run<NT, NV, NC, NM, 3>(d, WHAT); run<NT, NV, NC, NM, 4>(d, WHAT);

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  ALLW(8, 0, 0, 0, "8 transcendentals")
  ALLW(0, 16, 0, 0, "16 v_fma")
  ALLW(0, 0, 0, 5, "5 mfma 16x16x16 f16")
  ALLW(8, 14, 4, 5, "bwd tile: 8 tr + 14 fma + 4 cvt + 5 mfma")
  ALLW(8, 6, 2, 2, "fwd tile: 8 tr + 6 fma + 2 cvt + 2 mfma")
  ALLW(4, 14, 4, 5, "bwd tile with 4 transcendentals")
  run<8, 14, 4, 5, 2, true>(d, "bwd tile, the 14 fma as 7 v_pk_fma");
  run<0, 16, 0, 0, 2, true>(d, "16 fma as 8 v_pk_fma");
  run<0, 16, 0, 5, 2, false>(d, "16 v_fma + 5 mfma");
  run<0, 16, 0, 5, 2, true>(d, "8 v_pk_fma + 5 mfma");
  hipFree(d);
  return 0;
}
