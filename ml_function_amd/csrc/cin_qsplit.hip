// Instantiations + dispatch of the split-bf16 kernels of the merged quadratic tail (cin_qsplit.h).
#include "cin_qsplit.h"

namespace fil {

bool cin_launch_fwdq_b(hipStream_t st, int JT, const float* x2T, int XL, const u32x4* Wb, int NT, const float* bias1, const float* wsn, int JTG,
                       const float* cvec, float* x1T, float* RT, int HS, float* pool1, float* pool_p, float* pool_L, int M, int F, int H, CinHeadFold hf) {
  const size_t sh = (size_t)kQsStages * kQsStageBytes;
#define FIL_FQB(JTV)                                                                                                                                  \
  case JTV:                                                                                                                                           \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_fwdq_b_kernel<JTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);            \
    hipLaunchKernelGGL((cin_fwdq_b_kernel<JTV>), dim3(cdiv(M, 256)), dim3(512), sh, st, x2T, XL, Wb, NT, bias1, wsn, JTG, cvec, x1T, RT, HS, pool1, pool_p, \
                       pool_L, M, F, H, hf);                                                                                                          \
    break;
  switch (JT) { FIL_FQB(10) default: return false; }
#undef FIL_FQB
  return true;
}

}  // namespace fil
