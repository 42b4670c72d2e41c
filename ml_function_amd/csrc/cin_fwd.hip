// Instantiations + dispatch of cin_fwd3_kernel<MB, JT>.
#include "cin_kernels.h"
#include "cin_launch.h"

namespace fil {

template <int MB, int JT>
static void fwd3(hipStream_t st, dim3 grid, const float* xT, const float* xpT, int xps, const float* Wf, const float* bias,
                 float* xoutT, int HS, float* pool_part, int M, int F, int Hp, int H) {
  hipLaunchKernelGGL((cin_fwd3_kernel<MB, JT>), grid, dim3(kCinThreads), 0, st, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H);
}

void cin_launch_fwd3(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* xpT, int xps, const float* Wf,
                     const float* bias, float* xoutT, int HS, float* pool_part, int M, int F, int Hp, int H) {
#define FIL_F3(JTV)                                                                                  \
  case JTV:                                                                                          \
    if (MB == 2) fwd3<2, JTV>(st, grid, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H);  \
    else fwd3<1, JTV>(st, grid, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H);          \
    break;
  switch (JT) { FIL_F3(4) FIL_F3(8) FIL_F3(12) FIL_F3(16) FIL_F3(20) FIL_F3(24) FIL_F3(28) FIL_F3(32) }
#undef FIL_F3
}

template <int MB, int JT>
static void fwd3s(hipStream_t st, dim3 grid, const float* xT, const float* Wf, const float* bias, float* xoutT, int HS, float* pool_part,
                  int M, int F, int H) {
  hipLaunchKernelGGL((cin_fwd3_kernel<MB, JT, true>), grid, dim3(kCinThreads), 0, st, xT, xT, F, Wf, bias, xoutT, HS, pool_part, M, F, F, H);
}

void cin_launch_fwd3_sym(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* Wf, const float* bias, float* xoutT,
                         int HS, float* pool_part, int M, int F, int H) {
#define FIL_F3S(JTV)                                                                  \
  case JTV:                                                                           \
    if (MB == 2) fwd3s<2, JTV>(st, grid, xT, Wf, bias, xoutT, HS, pool_part, M, F, H); \
    else fwd3s<1, JTV>(st, grid, xT, Wf, bias, xoutT, HS, pool_part, M, F, H);         \
    break;
  switch (JT) { FIL_F3S(2) FIL_F3S(4) FIL_F3S(6) FIL_F3S(8) FIL_F3S(10) FIL_F3S(12) FIL_F3S(14) FIL_F3S(16) FIL_F3S(18) }
#undef FIL_F3S
}

}  // namespace fil
