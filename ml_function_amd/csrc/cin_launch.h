// Launchers of the templated CIN GEMM kernels (instantiated in separate translation units so they compile in parallel).
#pragma once
#include <hip/hip_runtime.h>

namespace fil {

// JT = steps per h of the streaming kernels: ceil(F/2) rounded up to a multiple of 4 (menu 4..32, F <= 64)
inline int cin_jt_of(int F) { return ((F + 1) / 2 + 3) / 4 * 4; }

// steps per h of the symmetric first-layer kernels: d = 0..F/2 in pairs, rounded up to an even count (menu 2..18)
constexpr int kCinMaxFields = 64;   // check_shape's limit; the JT menus of the launchers are sized for it
constexpr int cin_jt_sym(int F) { return ((F / 2 + 1 + 1) / 2 + 1) / 2 * 2; }
static_assert(cin_jt_sym(kCinMaxFields) <= 18, "the symmetric kernels' JT menu (2..18) must cover every legal field count");

// ks = 4: four waves share a block of 32 rows and split the reduction over h (small M; MB = 1 only; the grid must
// then be cdiv(M, 32) workgroups in x)
// x2T [M][XL]: the wrapped rows of x (cin_transpose_in_body), XL = cin_x2_len(F)
inline int cin_x2_len(int F) { return F + 2 * cin_jt_sym(F); }
void cin_launch_fwd3_sym(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* x2T, int XL, const float* Wf, const float* bias,
                         float* xoutT, int HS, float* pool_part, int M, int F, int H, int ks = 1);

// wsn != nullptr: also sum-pool the next (last, mode 0) layer in the epilogue -> pool_next (see cin_fwd3_kernel)
void cin_launch_fwd3(hipStream_t st, int MB, int JT, dim3 grid, const float* xT, const float* xpT, int xps, const float* Wf,
                     const float* bias, float* xoutT, int HS, float* pool_part, int M, int F, int Hp, int H,
                     const float* wsn = nullptr, const float* bias_next = nullptr, int H_next = 0, float* pool_next = nullptr);

void cin_launch_dz3(hipStream_t st, int MB, int JT, int NHMAX, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT,
                    const float* xpT, int xps, const float* dPprev, int ldp, int K, float* GprevT, int HSp, float* gx0T, float* dxT,
                    int accumulate, int M, int F, int Hp, int H, int periods);

// MFMA data gradients of the last layer in mode 0 (L >= 2): see cin_last_bwd2_kernel
void cin_launch_last_bwd2(hipStream_t st, int JT, const float* xT, const float* xpT, int xps, const float* wsum, const float* wsn,
                          const float* dP, int ldp, const float* dPprev, float* GprevT, int HSp, float* dxT, int M, int F, int K, int Hp,
                          const float* Radd = nullptr, int HSr = 0, const float* dPadd = nullptr, float* colpart = nullptr);

// symmetric first layer (x^{l-1} = x); FR = field rows of the LDS scratch (see cin_dz_sym_rows)
void cin_launch_dz3_sym(hipStream_t st, int MB, int JT, int NHMAX, dim3 grid, const float* gT, int HS, const float* Wz, const float* xT,
                        float* gx0T, float* dxT, int accumulate, int M, int F, int H, int periods, int ks = 1);

// tiles per period / h per period of the dZ kernel for a given JT (mirrors the constexprs in cin_dz3_kernel)
inline int cin_gcd(int a, int b) { return b == 0 ? a : cin_gcd(b, a % b); }
inline int cin_dz_tiles_per_period(int JT) { return JT / cin_gcd(16, JT); }
inline int cin_dz_h_per_period(int JT) { return 16 * cin_dz_tiles_per_period(JT) / JT; }

// LDS field rows of the symmetric dZ kernel: h + 2j + half stays below F + rows for every (padded) slot
inline int cin_dz_sym_rows(int F, int JT) { return F > cin_dz_h_per_period(JT) + 2 * JT ? F : cin_dz_h_per_period(JT) + 2 * JT; }

// ---- fused tail (cin_tail.h): the last two layers through the pooled weights of the last one
// A / Y / Q carry F+2 columns (dP_p | dP_L x_f | dP_L) in 16-column blocks; the forward walks the fields 4 per step
inline bool cin_tail_supported(int F) { return F >= 1 && F <= 62; }           // <= 4 column blocks (one float4 per lane and step)
inline int cin_tail_ncb(int F) { return (F + 2 + 15) / 16; }
inline int cin_tail_jt4(int F) {                                               // steps per h, on the kernel menu
  const int v = (F + 3) / 4;
  return v == 11 ? 12 : (v == 13 ? 14 : v);
}
inline int cin_tail_nq(int F) { return ((F + 2) / 2 + 3) / 4; }               // float4 per tile and lane of the dZ A stream
struct TailFwdArgs {
  const float *xT, *xpT;
  int xps;
  const float *Uf, *consts;
  float* Y;
  int JP;
  float *pool_p, *pool_L;
  int M, F, Hp;
  int ks;   // 4: four waves share a row block and split the reduction over h (RB = 2 only)
};
void cin_launch_tail_fwd(hipStream_t st, int RB, int JT4, int NCB, const TailFwdArgs& a);
struct TailDwArgs {
  const float *Apk, *xT, *xpT;
  int xps;
  float* part;
  int M, F, Hp, JP, rows_per_split, blocks_x, items;
  bool settle;
};
void cin_launch_tail_dw(hipStream_t st, int NCB, const TailDwArgs& a);
struct TailDzArgs {
  const float *Uz, *xT, *xpT;
  int xps;
  const float* Y;
  int JP;
  const float* dP;
  int ldp, K, lp, lL;
  float* GprevT;
  int HSp;
  float* dxT;
  int M, F, Hp, periods;
  int smode;
  int ks;   // 4: four waves share a block of 32 rows and split the periods
};
void cin_launch_tail_dz(hipStream_t st, int JT, int NQ, const TailDzArgs& a);

}  // namespace fil
