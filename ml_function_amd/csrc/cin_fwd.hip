// Instantiations + dispatch of cin_fwd3_kernel<MB, JT>.
#include "cin_kernels.h"
#include "cin_launch.h"

namespace fil {

template <int MB, int JT>
static void fwd3(hipStream_t st, dim3 grid, const float* xT, const float* xpT, int xps, const float* Wf, const float* bias,
                 float* xoutT, int HS, float* pool_part, int M, int F, int Hp, int H, const float* wsn, const float* bias_next,
                 int H_next, float* pool_next) {
  hipLaunchKernelGGL((cin_fwd3_kernel<MB, JT>), grid, dim3(kCinThreads), 0, st, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H, wsn,
                     bias_next, H_next, pool_next);
}

void cin_launch_fwd3(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* xpT, int xps, const float* Wf,
                     const float* bias, float* xoutT, int HS, float* pool_part, int M, int F, int Hp, int H, const float* wsn,
                     const float* bias_next, int H_next, float* pool_next) {
#define FIL_F3(JTV)                                                                                                                      \
  case JTV:                                                                                                                              \
    if (MB == 2) fwd3<2, JTV>(st, grid, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H, wsn, bias_next, H_next, pool_next);   \
    else fwd3<1, JTV>(st, grid, xT, xpT, xps, Wf, bias, xoutT, HS, pool_part, M, F, Hp, H, wsn, bias_next, H_next, pool_next);           \
    break;
  switch (JT) { FIL_F3(4) FIL_F3(8) FIL_F3(12) FIL_F3(16) FIL_F3(20) FIL_F3(24) FIL_F3(28) FIL_F3(32) }
#undef FIL_F3
}

template <int MB, int JT>
static void fwd3s(hipStream_t st, dim3 grid, const float* xT, const float* x2T, int XL, const float* Wf, const float* bias, float* xoutT, int HS,
                  float* pool_part, int M, int F, int H, int ks) {
  // the wrapped rows x2T [M][XL] travel in the xpT / xps slots (x^{l-1} = x needs no operand of its own)
  if constexpr (MB == 1) {
    if (ks == 4) {
      hipLaunchKernelGGL((cin_fwd3_kernel<1, JT, true, 4>), grid, dim3(kCinThreads), 0, st, xT, x2T, XL, Wf, bias, xoutT, HS, pool_part, M, F,
                         F, H, nullptr, nullptr, 0, nullptr);
      return;
    }
  }
  hipLaunchKernelGGL((cin_fwd3_kernel<MB, JT, true>), grid, dim3(kCinThreads), 0, st, xT, x2T, XL, Wf, bias, xoutT, HS, pool_part, M, F, F, H,
                     nullptr, nullptr, 0, nullptr);
}

void cin_launch_fwd3_sym(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* x2T, int XL, const float* Wf, const float* bias,
                         float* xoutT, int HS, float* pool_part, int M, int F, int H, int ks) {
#define FIL_F3S(JTV)                                                                             \
  case JTV:                                                                                      \
    if (MB == 2) fwd3s<2, JTV>(st, grid, xT, x2T, XL, Wf, bias, xoutT, HS, pool_part, M, F, H, 1);  \
    else fwd3s<1, JTV>(st, grid, xT, x2T, XL, Wf, bias, xoutT, HS, pool_part, M, F, H, ks);         \
    break;
  switch (JT) { FIL_F3S(2) FIL_F3S(4) FIL_F3S(6) FIL_F3S(8) FIL_F3S(10) FIL_F3S(12) FIL_F3S(14) FIL_F3S(16) FIL_F3S(18) }
#undef FIL_F3S
}

template <int JT>
static void last_bwd2(hipStream_t st, const float* xT, const float* xpT, int xps, const float* wsum, const float* wsn, const float* dP,
                      int ldp, const float* dPprev, float* GprevT, int HSp, float* dxT, int M, int F, int K, int Hp, const float* Radd, int HSr,
                      const float* dPadd, float* colpart) {
  const size_t sh = ((size_t)((Hp * F + 3) & ~3) + 4 * kLast2Stage) * sizeof(float);   // wsum | four waves' staging areas
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_last_bwd2_kernel<JT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_last_bwd2_kernel<JT>), dim3((M + 127) / 128), dim3(kCinThreads), sh, st, xT, xpT, xps, wsum, wsn, dP, ldp, dPprev,
                     GprevT, HSp, dxT, M, F, K, Hp, Radd, HSr, dPadd, colpart);
}

void cin_launch_last_bwd2(hipStream_t st, int JT, const float* xT, const float* xpT, int xps, const float* wsum, const float* wsn,
                          const float* dP, int ldp, const float* dPprev, float* GprevT, int HSp, float* dxT, int M, int F, int K, int Hp,
                          const float* Radd, int HSr, const float* dPadd, float* colpart) {
#define FIL_LB(JTV) \
  case JTV: last_bwd2<JTV>(st, xT, xpT, xps, wsum, wsn, dP, ldp, dPprev, GprevT, HSp, dxT, M, F, K, Hp, Radd, HSr, dPadd, colpart); break;
  switch (JT) { FIL_LB(4) FIL_LB(8) FIL_LB(12) FIL_LB(16) FIL_LB(20) FIL_LB(24) FIL_LB(28) FIL_LB(32) }
#undef FIL_LB
}

}  // namespace fil
