// Instantiations + dispatch of the fused-tail GEMM kernels (cin_tail.h).
#include "cin_tail.h"
#include "cin_launch.h"

namespace fil {

template <int RB, int JT4, int NCB>
static void tail_fwd(hipStream_t st, const TailFwdArgs& a) {
  if constexpr (RB == 2) {
    if (a.ks == 4) {   // one row block per workgroup, its four waves split the reduction over h
      hipLaunchKernelGGL((cin_tail_fwd_kernel<2, JT4, NCB, 4>), dim3(cdiv(a.M, 32)), dim3(kCinThreads), 0, st, a.xT, a.xpT, a.xps, a.Uf, a.consts,
                         a.Y, a.JP, a.pool_p, a.pool_L, a.M, a.F, a.Hp);
      return;
    }
  }
  const dim3 grid(cdiv(a.M, 64 * RB));
  hipLaunchKernelGGL((cin_tail_fwd_kernel<RB, JT4, NCB>), grid, dim3(kCinThreads), 0, st, a.xT, a.xpT, a.xps, a.Uf, a.consts, a.Y, a.JP,
                     a.pool_p, a.pool_L, a.M, a.F, a.Hp);
}

void cin_launch_tail_fwd(hipStream_t st, int RB, int JT4, int NCB, const TailFwdArgs& a) {
#define FIL_TF(J, N)                                  \
  if (JT4 == J && NCB == N) {                         \
    if (RB == 4) tail_fwd<4, J, N>(st, a);            \
    else tail_fwd<2, J, N>(st, a);                    \
    return;                                           \
  }
  FIL_TF(1, 1) FIL_TF(2, 1) FIL_TF(3, 1) FIL_TF(4, 1) FIL_TF(4, 2) FIL_TF(5, 2) FIL_TF(6, 2) FIL_TF(7, 2) FIL_TF(8, 2) FIL_TF(8, 3)
  FIL_TF(9, 3) FIL_TF(10, 3) FIL_TF(12, 3) FIL_TF(12, 4) FIL_TF(14, 4) FIL_TF(15, 4) FIL_TF(16, 4)
#undef FIL_TF
}

void cin_launch_tail_dw(hipStream_t st, int NCB, const TailDwArgs& a) {
  const dim3 grid((a.items + 7) / 8 * 8);
#define FIL_TW(N)                                                                                                                              \
  case N:                                                                                                                                      \
    if (a.settle)                                                                                                                              \
      hipLaunchKernelGGL((cin_tail_dw_kernel<N, true>), grid, dim3(kCinThreads), 0, st, a.Apk, a.xT, a.xpT, a.xps, a.part, a.M, a.F, a.Hp,      \
                         a.JP, a.rows_per_split, a.blocks_x, a.items);                                                                         \
    else                                                                                                                                       \
      hipLaunchKernelGGL((cin_tail_dw_kernel<N, false>), grid, dim3(kCinThreads), 0, st, a.Apk, a.xT, a.xpT, a.xps, a.part, a.M, a.F, a.Hp,     \
                         a.JP, a.rows_per_split, a.blocks_x, a.items);                                                                         \
    break;
  switch (NCB) { FIL_TW(1) FIL_TW(2) FIL_TW(3) FIL_TW(4) }
#undef FIL_TW
}

template <int JT, int NQ, int SMODE>
static void tail_dz(hipStream_t st, const TailDzArgs& a) {
  const size_t sh = ((size_t)JT * 256 + (size_t)4 * 32 * kGlStride) * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_tail_dz_kernel<JT, NQ, SMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_tail_dz_kernel<JT, NQ, SMODE>), dim3(cdiv(a.M, 128)), dim3(kCinThreads), sh, st, a.Uz, a.xT, a.xpT, a.xps, a.Y, a.JP, a.dP,
                     a.ldp, a.K, a.lp, a.lL, a.GprevT, a.HSp, a.dxT, a.M, a.F, a.Hp, a.periods);
}

template <int JT, int NQ>
static void tail_dz_ks4(hipStream_t st, const TailDzArgs& a) {
  const size_t sh = ((size_t)JT * 256 + (size_t)4 * 32 * kGlStride) * sizeof(float);
  if (sh > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cin_tail_dz_kernel<JT, NQ, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL((cin_tail_dz_kernel<JT, NQ, 0, 4>), dim3(cdiv(a.M, 32)), dim3(kCinThreads), sh, st, a.Uz, a.xT, a.xpT, a.xps, a.Y, a.JP, a.dP,
                     a.ldp, a.K, a.lp, a.lL, a.GprevT, a.HSp, a.dxT, a.M, a.F, a.Hp, a.periods);
}

void cin_launch_tail_dz(hipStream_t st, int JT, int NQ, const TailDzArgs& a) {
#define FIL_TZ(J)                                                                   \
  case J:                                                                           \
    if (a.ks == 4) {                                                                \
      if (NQ == J / 4) tail_dz_ks4<J, J / 4>(st, a);                                \
      else tail_dz_ks4<J, J / 4 + 1>(st, a);                                        \
    } else if (NQ == J / 4) tail_dz<J, J / 4, 0>(st, a);                            \
    else tail_dz<J, J / 4 + 1, 0>(st, a);                                           \
    break;
  if (a.smode != 0 && a.ks != 4 && JT == 20 && NQ == 5) {   // (experiment: slot placement variants at the north-star shape only)
    if (a.smode == 1) tail_dz<20, 5, 1>(st, a);
    else if (a.smode == 2) tail_dz<20, 5, 2>(st, a);
    else tail_dz<20, 5, 3>(st, a);
    return;
  }
  switch (JT) { FIL_TZ(4) FIL_TZ(8) FIL_TZ(12) FIL_TZ(16) FIL_TZ(20) FIL_TZ(24) FIL_TZ(28) FIL_TZ(32) }
#undef FIL_TZ
}

}  // namespace fil
