// Timing probe for cin_dw3_kernel variants (not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iml_function_amd/csrc tools/probe_dw.hip -o /tmp/probe_dw && /tmp/probe_dw
#include "cin_kernels.h"
#include <cstdio>
#include <vector>
using namespace fil;
namespace fil { void set_error(const char*, ...) {} int fail(int c, const char*, ...) { return c; } bool prof_enabled() { return false; } void prof_begin_scope(const char*, hipStream_t, double) {} void prof_end_scope(hipStream_t) {} }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main() {
  const int B = 4096, K = 16, F = 39, Hp = 128, H = 128, M = B * K, HS = 128, C = Hp * F;
  float *g, *x, *xp, *part;
  CHECK(hipMalloc(&g, (size_t)M * HS * 4)); CHECK(hipMalloc(&x, (size_t)M * F * 4)); CHECK(hipMalloc(&xp, (size_t)M * HS * 4));
  CHECK(hipMalloc(&part, (size_t)64 * C * H * 4));
  std::vector<float> h((size_t)M * HS);
  for (auto& v : h) v = (rand() / (float)RAND_MAX) - 0.5f;
  CHECK(hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(xp, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(x, h.data(), (size_t)M * F * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double flops = 2.0 * M * C * H;
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-40s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  };
  auto run = [&](auto kern, int MB, int depth, int splits, const char* name) {
    const int waves_c = (C + 32 * MB - 1) / (32 * MB), bx = (waves_c + 3) / 4;
    long rps = (M + splits - 1) / splits; rps = (rps + 2 * depth - 1) / (2 * depth) * (2 * depth);
    const int sp = (int)((M + rps - 1) / rps);
    char buf[128]; snprintf(buf, sizeof buf, "%s MB=%d D=%d splits=%d waves=%d", name, MB, depth, sp, waves_c * sp);
    timeit(buf, [&] { hipLaunchKernelGGL(kern, dim3(bx, sp, 1), dim3(256), 0, 0, g, HS, x, xp, HS, part, M, F, Hp, H, (int)rps); });
  };
  run(cin_dw3b_kernel<2, false, 8>, 2, 8, 13, "dw3b");
  run(cin_dw3b_kernel<2, false, 12>, 2, 12, 13, "dw3b");
  run(cin_dw3b_kernel<1, false, 8>, 1, 8, 26, "dw3b");
  run(cin_dw3b_kernel<1, false, 8>, 1, 8, 13, "dw3b");
  run(cin_dw3_kernel<2, true, 8>, 2, 8, 13, "dw3 XONES (no x loads)");
  run(cin_dw3_kernel<1, true, 8>, 1, 8, 26, "dw3 XONES (no x loads)");
  if (getenv("PD_MODE") && atoi(getenv("PD_MODE")) == 1) { run(cin_dw3_kernel<2, false, 8>, 2, 8, 13, "dw3"); return 0; }
  if (getenv("PD_MODE") && atoi(getenv("PD_MODE")) == 2) { run(cin_dw3_kernel<1, false, 8>, 1, 8, 26, "dw3"); return 0; }
  run(cin_dw3a_kernel<2, false, 8>, 2, 8, 13, "dw3a");
  run(cin_dw3a_kernel<2, false, 12>, 2, 12, 13, "dw3a");
  run(cin_dw3a_kernel<1, false, 8>, 1, 8, 13, "dw3a");
  run(cin_dw3a_kernel<1, false, 16>, 1, 16, 13, "dw3a");
  run(cin_dw3a_kernel<1, false, 8>, 1, 8, 26, "dw3a");
  run(cin_dw3_kernel<2, false, 8>, 2, 8, 13, "dw3");
  run(cin_dw3_kernel<2, false, 16>, 2, 16, 13, "dw3");
  run(cin_dw3_kernel<2, false, 4>, 2, 4, 13, "dw3");
  run(cin_dw3_kernel<1, false, 8>, 1, 8, 13, "dw3");
  run(cin_dw3_kernel<1, false, 16>, 1, 16, 13, "dw3");
  run(cin_dw3_kernel<1, false, 8>, 1, 8, 26, "dw3");
  run(cin_dw3_kernel<2, false, 8>, 2, 8, 26, "dw3");
  run(cin_dw3_kernel<2, false, 8>, 2, 8, 6, "dw3");
  return 0;
}
