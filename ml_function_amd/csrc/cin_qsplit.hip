// Instantiations + dispatch of the split-bf16 kernels of the merged quadratic tail (cin_qsplit.h).
#include "cin_qsplit.h"

namespace fil {

bool cin_launch_fwdq_b(hipStream_t st, int JT, const float* x2T, int XL, const u32x4* Wb, int NT, const float* bias1, const float* wsn, int JTG,
                       const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf) {
  // (4-wave workgroups, two per CU, each with a ring of three; one 8-wave workgroup per CU sharing a ring of four measured the same:
  // 127.7 against 129.7 us)
  const size_t sh = (size_t)3 * kQsStageBytes;
#define FIL_FQB(JTV)                                                                                                                                  \
  case JTV:                                                                                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwdq_b_kernel<JTV, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);         \
    hipLaunchKernelGGL((cin_fwdq_b_kernel<JTV, 4>), dim3(cdiv(M, 128)), dim3(256), sh, st, x2T, XL, Wb, NT, bias1, wsn, JTG, cvec, x1T, RT, HS, pool1,    \
                       pool_p, pool_L, M, F, H, hf);                                                                                                  \
    break;
  switch (JT) { FIL_FQB(2) FIL_FQB(4) FIL_FQB(6) FIL_FQB(8) FIL_FQB(10) FIL_FQB(12) default: return false; }
#undef FIL_FQB
  return true;
}

bool cin_launch_dz2_b(hipStream_t st, int JT, const float* g1T, const float* g2T, int HS, const float* dsc, int ldp, int K, const u32x4* Wzb1,
                      const u32x4* Wzb2, const float* xT, float* dxT, int accumulate, int M, int F, int H1, int H2, int periods, float* dx,
                      const float* cvec) {
  // (blocks of four slots only: F >= HPP + 2 JT, as cin_launch_dz2 picks them; other shapes keep the exact kernel)
  if (!(JT >= 4 && F >= cin_dz_h_per_period(JT) + 2 * JT)) return false;
  const int FR = cin_dz2_rows(F, JT);
  const size_t sh = (size_t)FR * kDz2FieldStride * sizeof(float);
#define FIL_Z2B(JTV)                                                                                                                           \
  case JTV:                                                                                                                                    \
    if (sh > 48 * 1024)                                                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dz2_b_kernel<JTV, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    hipLaunchKernelGGL((cin_dz2_b_kernel<JTV, 4>), dim3(cdiv(M, 128)), dim3(kCinThreads), sh, st, g1T, g2T, HS, dsc, ldp, K, Wzb1, Wzb2, xT, dxT,  \
                       accumulate, M, F, H1, H2, periods, FR, dx, cvec);                                                                       \
    break;
  switch (JT) { FIL_Z2B(4) FIL_Z2B(6) FIL_Z2B(8) FIL_Z2B(10) FIL_Z2B(12) default: return false; }
#undef FIL_Z2B
  return true;
}

void cin_launch_dwq_b(hipStream_t st, const DwqbPlan& p, const float* gT, const float* x1T, int HS, const float* xe, int XE, float* part, int M, int F,
                      int symD) {
  const size_t sh = (size_t)kDwqbHalfBytes;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_dwq_b_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL(cin_dwq_b_kernel<3>, dim3((p.wgs + 7) / 8 * 8), dim3(256), sh, st, gT, x1T, HS, xe, XE, part, M, F, symD, p.rows_per_split, p.splits,
                     p.groups, p.items);
}

}  // namespace fil
