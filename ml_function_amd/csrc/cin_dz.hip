// Instantiations + dispatch of cin_dz3_kernel<MB, JT, NHMAX>.
#include "cin_kernels.h"
#include "cin_launch.h"

namespace fil {

template <int MB, int JT, int NHMAX>
static void dz3(hipStream_t st, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT, const float* xpT, int xps,
                const float* dPprev, int ldp, int K, float* GprevT, int HSp, float* gx0T, float* dxT, int accumulate, int M, int F,
                int Hp, int H, int periods) {
  // per-lane LDS scratch: x fragment + dX accumulators, [2][MB][JT][256] floats, + the G^{l-1} line buffers [4 waves][MB][32][kGlStride]
  const size_t sh = ((size_t)2 * MB * JT * 256 + (size_t)4 * MB * 32 * kGlStride) * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz3_kernel<MB, JT, NHMAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_dz3_kernel<MB, JT, NHMAX>), grid, dim3(kCinThreads), sh, st, gT, HS, Wz, xT, xpT, xps, dPprev, ldp, K,
                     GprevT, HSp, gx0T, dxT, accumulate, M, F, Hp, H, periods, 0);
}

void cin_launch_dz3(hipStream_t st, int MB, int JT, int NHMAX, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT,
                    const float* xpT, int xps, const float* dPprev, int ldp, int K, float* GprevT, int HSp, float* gx0T, float* dxT,
                    int accumulate, int M, int F, int Hp, int H, int periods) {
#define FIL_ARGS st, grid, gT, HS, Wz, xT, xpT, xps, dPprev, ldp, K, GprevT, HSp, gx0T, dxT, accumulate, M, F, Hp, H, periods
#define FIL_Z3(JTV)                                                                                 \
  case JTV:                                                                                         \
    if (NHMAX == 128) dz3<1, JTV, 128>(FIL_ARGS);                                                   \
    else if (MB == 2) dz3<2, JTV, 64>(FIL_ARGS);                                                    \
    else dz3<1, JTV, 64>(FIL_ARGS);                                                                 \
    break;
  switch (JT) { FIL_Z3(4) FIL_Z3(8) FIL_Z3(12) FIL_Z3(16) FIL_Z3(20) FIL_Z3(24) FIL_Z3(28) FIL_Z3(32) }
#undef FIL_Z3
#undef FIL_ARGS
}

template <int MB, int JT, int NHMAX>
static void dz3s(hipStream_t st, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT, float* gx0T, float* dxT,
                 int accumulate, int M, int F, int H, int periods, int ks = 1) {
  const int FR = cin_dz_sym_rows(F, JT);
  const size_t sh = (size_t)2 * MB * FR * kSymStride * sizeof(float);
  if constexpr (MB == 1 && NHMAX == 64) {
    if (ks == 4) {
      if (sh > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz3_kernel<1, JT, 64, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipLaunchKernelGGL((cin_dz3_kernel<1, JT, 64, true, 4>), grid, dim3(kCinThreads), sh, st, gT, HS, Wz, xT, xT, F, nullptr, 0, 1, nullptr, 0,
                         gx0T, dxT, accumulate, M, F, F, H, periods, FR);
      return;
    }
  }
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz3_kernel<MB, JT, NHMAX, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_dz3_kernel<MB, JT, NHMAX, true>), grid, dim3(kCinThreads), sh, st, gT, HS, Wz, xT, xT, F, nullptr, 0, 1, nullptr, 0,
                     gx0T, dxT, accumulate, M, F, F, H, periods, FR);
}

void cin_launch_dz3_sym(hipStream_t st, int MB, int JT, int NHMAX, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT,
                        float* gx0T, float* dxT, int accumulate, int M, int F, int H, int periods, int ks) {
#define FIL_ARGS st, grid, gT, HS, Wz, xT, gx0T, dxT, accumulate, M, F, H, periods
#define FIL_Z3S(JTV)                                                                                              \
  case JTV:                                                                                                       \
    if (NHMAX == 128) dz3s<1, JTV, 128>(FIL_ARGS);                                                                \
    else if (MB == 2) dz3s<2, JTV, 64>(FIL_ARGS);                                                                 \
    else dz3s<1, JTV, 64>(FIL_ARGS, ks);                                                                          \
    break;
  switch (JT) { FIL_Z3S(2) FIL_Z3S(4) FIL_Z3S(6) FIL_Z3S(8) FIL_Z3S(10) FIL_Z3S(12) FIL_Z3S(14) FIL_Z3S(16) FIL_Z3S(18) }
#undef FIL_Z3S
#undef FIL_ARGS
}

}  // namespace fil
