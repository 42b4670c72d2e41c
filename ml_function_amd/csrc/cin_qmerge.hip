// Instantiations + dispatch of cin_dz2_kernel<JT> (both data-gradient passes of the quadratic tail in one launch) and
// cin_fwdq_kernel<JT> (first layer + quadratic form forward, 256 columns, all three sum-pools).
#include "cin_qmerge.h"

namespace fil {

template <int JT, int G>
static void dz2g(hipStream_t st, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1, const float* Wz2,
                 const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods, float* dx, const float* cvec) {
  const int FR = cin_dz2_rows(F, JT);
  const size_t sh = (size_t)FR * kDz2FieldStride * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz2_kernel<JT, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_dz2_kernel<JT, G>), dim3(cdiv(M, 128)), dim3(kCinThreads), sh, st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1,
                     H2, periods, FR, dx, cvec);
}
template <int JT>
static void dz2(hipStream_t st, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1, const float* Wz2,
                const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods, float* dx, const float* cvec) {
  // slots per block (cin_dz2_kernel): four where no two slots of a block can touch one word, else one at a time
  if (JT >= 4 && F >= cin_dz_h_per_period(JT) + 2 * JT) dz2g<JT, (JT >= 4 ? 4 : 1)>(st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1, H2, periods, dx, cvec);
  else dz2g<JT, 1>(st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1, H2, periods, dx, cvec);
}

bool cin_launch_dz2(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const float* Wz1,
                    const float* Wz2, const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods, float* dx, const float* cvec) {
#define FIL_Z2(JTV) \
  case JTV: dz2<JTV>(st, g1T, g2T, HS, dsc, ldp, K, Wz1, Wz2, xT, dxT, accumulate, M, F, H1, H2, periods, dx, cvec); break;
  // (an unlisted JT must not look like a launch: the caller would mark dx / the head as done)
  switch (JT) { FIL_Z2(2) FIL_Z2(4) FIL_Z2(6) FIL_Z2(8) FIL_Z2(10) FIL_Z2(12) FIL_Z2(14) FIL_Z2(16) FIL_Z2(18) default: return false; }
#undef FIL_Z2
  return true;
}

bool cin_launch_fwdq(hipStream_t st, int JT, const float* x2T, int XL, const float* W1f, const float* WTf, const float* bias1, const float* wsn, int JTG,
                     const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf) {
#define FIL_FQ(JTV)                                                                                                                             \
  case JTV:                                                                                                                                     \
    hipLaunchKernelGGL((cin_fwdq_kernel<JTV>), dim3(cdiv(M, 128)), dim3(kCinThreads), 0, st, x2T, XL, W1f, WTf, bias1, wsn, JTG, cvec, x1T, RT, HS, \
                       pool1, pool_p, pool_L, M, F, H, hf);                                                                                       \
    break;
  switch (JT) { FIL_FQ(2) FIL_FQ(4) FIL_FQ(6) FIL_FQ(8) FIL_FQ(10) FIL_FQ(12) FIL_FQ(14) FIL_FQ(16) FIL_FQ(18) default: return false; }
#undef FIL_FQ
  return true;
}

}  // namespace fil
