// Instantiations + dispatch of cin_dzq_kernel<MB, JT, KH> (merged data gradients of the quadratic tail).
#include "cin_qmerge.h"

namespace fil {

template <int JT>
static void dzq(hipStream_t st, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wzq, const float* xT,
                float* dxT, int accumulate, int M, int F, int H1, int H2, int periods) {
  constexpr int MB = 2;
  const int FR = cin_dzq_rows(F, JT);
  const size_t sh = (size_t)FR * MB * kDzqFieldStride * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dzq_kernel<MB, JT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_dzq_kernel<MB, JT, 2>), dim3(cdiv(M, 128 * MB)), dim3(kCinThreads), sh, st, g1T, g2T, HS, dsc, ldp, K, Wzq, xT, dxT, accumulate,
                     M, F, H1, H2, periods, FR);
}

void cin_launch_dzq(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wzq,
                    const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods) {
#define FIL_ZQ(JTV) \
  case JTV: dzq<JTV>(st, g1T, g2T, HS, dsc, ldp, K, Wzq, xT, dxT, accumulate, M, F, H1, H2, periods); break;
  switch (JT) { FIL_ZQ(2) FIL_ZQ(4) FIL_ZQ(6) FIL_ZQ(8) FIL_ZQ(10) FIL_ZQ(12) FIL_ZQ(14) FIL_ZQ(16) FIL_ZQ(18) }
#undef FIL_ZQ
}

}  // namespace fil
