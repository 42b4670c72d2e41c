// v_mfma_f32_16x16x4_f32 on gfx950: what does the issue pattern of the fused-tail kernels cost?  One wave per SIMD, NACC
// independent accumulators, per group of G MFMAs: one v_mul producing their shared A operand (MUL), B operands from registers.
// Compared with the 32x32x2 form at equal flops.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_mfma16.hip -o /tmp/probe_mfma16 && /tmp/probe_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC accumulators, steps of RB groups x G MFMAs; MUL: a = x[rb] * y per group (fresh A operand per group)
template <int RB, int G, bool MUL, bool SAMEB>
__global__ __launch_bounds__(256, 1) void probe16(float* out, int iters, float seed) {
  f32x4 acc[RB][G];
  for (int i = 0; i < RB; ++i)
    for (int g = 0; g < G; ++g)
      for (int e = 0; e < 4; ++e) acc[i][g][e] = 0.f;
  float x[RB], b[G];
  for (int i = 0; i < RB; ++i) x[i] = seed + i + threadIdx.x;
  for (int g = 0; g < G; ++g) b[g] = seed * 0.25f + g;
  float y = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float a = MUL ? x[rb] * y : x[rb];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, SAMEB ? b[0] : b[g], acc[rb][g], 0, 0, 0);
      }
      y += 1.0f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = y;
  for (int i = 0; i < RB; ++i)
    for (int g = 0; g < G; ++g)
      for (int e = 0; e < 4; ++e) s += acc[i][g][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same 4 x 3 pattern with the A operands of step u+1 computed during step u (one v_mul between the MFMAs of a row block,
// consumed 12 MFMAs later): no VALU-write -> MFMA-read wait states in front of any MFMA
template <int RB, int G, int SPREAD>
__global__ __launch_bounds__(256, 1) void probe16p(float* out, int iters, float seed) {
  f32x4 acc[RB][G];
  for (int i = 0; i < RB; ++i)
    for (int g = 0; g < G; ++g)
      for (int e = 0; e < 4; ++e) acc[i][g][e] = 0.f;
  float x[RB], b[G], a[RB], an[RB];
  for (int i = 0; i < RB; ++i) x[i] = seed + i + threadIdx.x;
  for (int g = 0; g < G; ++g) b[g] = seed * 0.25f + g;
  float y = seed;
  for (int i = 0; i < RB; ++i) a[i] = x[i] * y;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      y += 1.0f;
      if (SPREAD == 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) an[rb] = x[rb] * y;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int g = 0; g < G; ++g) acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb], b[g], acc[rb][g], 0, 0, 0);
        if (SPREAD == 1) {
          an[rb] = x[rb] * y;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
    }
  }
  float s = y;
  for (int i = 0; i < RB; ++i)
    for (int g = 0; g < G; ++g)
      for (int e = 0; e < 4; ++e) s += acc[i][g][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool MUL>
__global__ __launch_bounds__(256, 1) void probe32(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float x = seed + threadIdx.x, b = seed * 0.25f, y = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = MUL ? x * y : x;
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + i, acc[i], 0, 0, 0);
      y += 1.0f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = y;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void timeit(const char* what, K launch, double flops) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.3f ms  %6.1f TFLOP/s\n", what, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  const int iters = 5000;
  const double w = 256.0 * 4;  // waves
#define P16(RB, G, MUL, SB, txt) \
  timeit(txt, [&] { probe16<RB, G, MUL, SB><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * RB * G * 2048.0)
  P16(4, 3, false, false, "16x16x4: 4 row blocks x 3 col blocks, A from registers");
  P16(4, 3, true, false, "16x16x4: 4 x 3, one v_mul per row block (fused-tail fwd)");
  P16(4, 4, true, false, "16x16x4: 4 x 4, one v_mul per row block");
  P16(2, 3, true, false, "16x16x4: 2 x 3, one v_mul per row block");
  P16(8, 3, true, false, "16x16x4: 8 x 3, one v_mul per row block");
  P16(4, 3, true, true, "16x16x4: 4 x 3, v_mul, same B register for the 3");
  P16(1, 12, true, false, "16x16x4: 1 x 12, one v_mul per 12 MFMAs");
  timeit("16x16x4: 4 x 3, next step's 4 v_mul in a block up front", [&] { probe16p<4, 3, 0><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * 12 * 2048.0);
  timeit("16x16x4: 4 x 3, next step's v_mul spread between row blocks", [&] { probe16p<4, 3, 1><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * 12 * 2048.0);
  timeit("32x32x2: 4 accumulators, A from registers", [&] { probe32<4, false><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * 4 * 4096.0);
  timeit("32x32x2: 4 accumulators, one v_mul per 4 MFMAs", [&] { probe32<4, true><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * 4 * 4096.0);
  timeit("32x32x2: 2 accumulators, one v_mul per 2 MFMAs", [&] { probe32<2, true><<<256, 256>>>(d, iters, 1.0f); }, w * iters * 4.0 * 2 * 4096.0);
  return 0;
}
