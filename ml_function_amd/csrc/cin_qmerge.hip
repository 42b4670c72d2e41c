// Instantiations + dispatch of cin_dz2_kernel<JT> (both data-gradient passes of the quadratic tail in one launch).
#include "cin_qmerge.h"

namespace fil {

template <int JT>
static void dz2(hipStream_t st, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1, const float* Wz2,
                const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods) {
  const int FR = cin_dz2_rows(F, JT);
  const size_t sh = (size_t)FR * kDz2FieldStride * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz2_kernel<JT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_dz2_kernel<JT>), dim3(cdiv(M, 128)), dim3(kCinThreads), sh, st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1,
                     H2, periods, FR);
}

void cin_launch_dz2(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1,
                    const float* Wz2, const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods) {
#define FIL_Z2(JTV) \
  case JTV: dz2<JTV>(st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1, H2, periods); break;
  switch (JT) { FIL_Z2(2) FIL_Z2(4) FIL_Z2(6) FIL_Z2(8) FIL_Z2(10) FIL_Z2(12) FIL_Z2(14) FIL_Z2(16) FIL_Z2(18) }
#undef FIL_Z2
}

}  // namespace fil
