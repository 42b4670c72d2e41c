// Split-bf16 ("bf16x3") building blocks shared by the small kernels that pack the weight planes (cin_qtail.h, cin_qmerge.h) and by the
// GEMM kernels that consume them (cin_qsplit.h, where the mode is described).
#pragma once
#include "cin_kernels.h"
#include "cin_launch.h"

namespace fil {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32b(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A B to fp32 accuracy from the pieces, small terms first
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  c = mfma32b(a[2], b[0], c);
  c = mfma32b(a[0], b[2], c);
  c = mfma32b(a[1], b[1], c);
  c = mfma32b(a[1], b[0], c);
  c = mfma32b(a[0], b[1], c);
  c = mfma32b(a[0], b[0], c);
  return c;
}

// two fp32 values -> their top halves in one dword (element 0 in the low half): a bf16 pair by truncation
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ float top8(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
// eight fp32 values -> the three bf16 planes of one MFMA operand (6.5 vector instructions per value: and, sub, and, sub + 1.5 perm)
__device__ __forceinline__ void split3(const float (&p)[8], u32x4 (&a)[3]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float p0 = p[2 * q], p1 = p[2 * q + 1];
    const float r0 = p0 - top8(p0), r1 = p1 - top8(p1);
    const float s0 = r0 - top8(r0), s1 = r1 - top8(r1);
    a[0][q] = pack_hi(p0, p1);
    a[1][q] = pack_hi(r0, r1);
    a[2][q] = pack_hi(s0, s1);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Geometry of the pair slots.  A wave half walks the slots s = h JT + j (the exact kernels' steps: pair (h, (h + 2j + half) mod F)) eight
// per MFMA; a PERIOD is HPS values of h = KP whole MFMA steps (HPS JT = 8 KP, KP >= 2), inside which slot -> (h - h0, j) is compile-time.
template <int JT>
struct QsGeo {
  static constexpr int HPS0 = 8 / gcd_c(8, JT);
  static constexpr int HPS = HPS0 * JT / 8 >= 2 ? HPS0 : 2 * HPS0;
  static constexpr int KP = HPS * JT / 8;
  static constexpr int WS = HPS + 2 * JT - 2;   // window of wrapped positions a period touches: t = (h - h0) + 2j
  static_assert(HPS * JT == 8 * KP && KP >= 2, "a period is a whole number (>= 2) of 8-slot steps");
};
inline int cin_qs_hps(int JT) {
  const int h0 = 8 / cin_gcd(8, JT);
  return h0 * JT / 8 >= 2 ? h0 : 2 * h0;
}
inline int cin_qs_steps(int F, int JT) {   // MFMA steps of the forward's reduction (slots past F JT carry zero weights)
  const int hps = cin_qs_hps(JT);
  return (F + hps - 1) / hps * (hps * JT / 8);
}
constexpr int kQsStageBytes = 24 * 1024;   // one forward step of B planes: [plane 3][column block 8][lane 64][8 bf16]
constexpr int kQsStages = 4;

// pair weight of slot (h, d) of a pair-symmetric layer (cin_pack_wf_sym_body's rule): W[(h,f)] + W[(f,h)], f = (h + d) mod F; the
// diagonal once; half the sum where an even F meets a pair from both ends; zero past d = F/2, h = F - 1 and column H - 1
__device__ __forceinline__ float cin_sym_weight(const float* __restrict__ W, int F, int H, int h, int d, int col) {
  if (h >= F || d > F / 2 || col >= H) return 0.f;
  const int f = (h + d) % F;
  if (d == 0) return W[((long)h * F + h) * H + col];
  const float v = W[((long)h * F + f) * H + col] + W[((long)f * F + h) * H + col];
  return 2 * d == F ? 0.5f * v : v;
}

// Forward planes Wb [step t][plane][nb 0..7][lane][8 bf16]: element e of lane (r, half) = the pair weight of slot s = 8 t + e (h = s / JT,
// d = 2 (s mod JT) + half), column 4 r + (nb & 3) of W1 [F*F][H1] (nb < 4) or of T [F*F][HT] (nb >= 4).  One thread per (t, nb, lane);
// rides in cin_qtail_pack_kernel's launch (T is complete there).
__device__ __forceinline__ void cin_qs_pack_wb_body(const float* __restrict__ W1, int H1, const float* __restrict__ T, int HT, u32x4* __restrict__ Wb,
                                                    int NT, int F, int JT, int bid, int nblocks) {
  for (int idx = bid * 256 + threadIdx.x; idx < NT * 512; idx += nblocks * 256) {
    const int lane = idx & 63, nb = (idx >> 6) & 7, t = idx >> 9;
    const int r = lane & 31, half = lane >> 5;
    const float* W = nb < 4 ? W1 : T;
    const int H = nb < 4 ? H1 : HT, col = 4 * r + (nb & 3);
    float p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int s = 8 * t + e, h = s / JT;
      p[e] = cin_sym_weight(W, F, H, h, 2 * (s - h * JT) + half, col);
    }
    u32x4 a[3];
    split3(p, a);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) Wb[((long)(t * 3 + pl) * 8 + nb) * 64 + lane] = a[pl];
  }
}

// Data-gradient planes Wzb [tile][step t][plane][lane][8 bf16]: element e of lane (r, half) = the pair weight of slot row r of the tile
// (cin_pack_wz_sym_body's slot order: rr = (r & 3) + 4 (r >> 3), parity (r >> 2) & 1, slot 16 tile + rr), column half*64 + 8 t + e.
// One thread per (tile, t, lane); rides in cin_qtail_xe_kernel's launch.
__device__ __forceinline__ void cin_qs_pack_wz_body(const float* __restrict__ W, int H, u32x4* __restrict__ Wzb, int tiles, int F, int JT, int bid,
                                                    int nblocks) {
  for (int idx = bid * 256 + threadIdx.x; idx < tiles * 512; idx += nblocks * 256) {
    const int lane = idx & 63, t = (idx >> 6) & 7, tile = idx >> 9;
    const int r = lane & 31, half = lane >> 5;
    const int rr = (r & 3) + 4 * (r >> 3), hf = (r >> 2) & 1;
    const int slot = 16 * tile + rr, h = slot / JT, d = 2 * (slot - h * JT) + hf;
    float p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) p[e] = cin_sym_weight(W, F, H, h, d, half * 64 + 8 * t + e);
    u32x4 a[3];
    split3(p, a);
    u32x4* dst = Wzb + ((long)(tile * 8 + t) * 3) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) dst[pl * 64] = a[pl];
  }
}

}  // namespace fil
