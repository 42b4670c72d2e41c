// Shared host/device helpers for libfil_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/fil.h"

namespace fil {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define FIL_CHECK_ARG(cond)                                                              \
  do {                                                                                   \
    if (!(cond)) return ::fil::fail(FIL_ERR_ARG, "%s: bad argument: %s", __func__, #cond); \
  } while (0)

#define FIL_CHECK_LAUNCH()                                                                          \
  do {                                                                                              \
    hipError_t e__ = hipGetLastError();                                                             \
    if (e__ != hipSuccess)                                                                          \
      return ::fil::fail(FIL_ERR_HIP, "%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Opt-in per-kernel timing (fil_profile_begin/_end in include/fil.h): when enabled, every major launch is
// bracketed by a pair of HIP events recorded on the launch stream.  Off by default (zero overhead, capture-safe).
bool prof_enabled();
bool prof_begin_scope(const char* name, hipStream_t st, double work, double executed);
void prof_end_scope(hipStream_t st);
// roctx ranges around the same scopes (host-side push/pop; `rocprofv3 --marker-trace --kernel-trace` shows which launches
// belong to which step of the layer).  Off unless FIL_ROCTX=1 was set when the library was loaded (read once).
bool roctx_enabled();
void roctx_push(const char* name);
void roctx_pop();
struct ProfScope {
  hipStream_t st;
  bool on, rx;
  // work = algorithmic flops (MFMA kernels) or bytes (streaming kernels) of this launch -- what the REFERENCE graph spends on
  // the step this scope stands for; executed = what the kernels inside the scope really compute (< work where an exact algebraic
  // restructuring -- pair symmetry, pooled-weight contraction -- removes products; default: the same).  Both reported verbatim.
  ProfScope(const char* name, hipStream_t s, double work = 0.0, double executed = -1.0) : st(s), on(prof_enabled()), rx(roctx_enabled()) {
    if (rx) roctx_push(name);
    if (on) on = prof_begin_scope(name, st, work, executed < 0.0 ? work : executed);
  }
  ~ProfScope() {
    if (on) prof_end_scope(st);
    if (rx) roctx_pop();
  }
};

// Carves consecutive 256-byte aligned regions out of a caller-provided workspace.
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(base + off);
    off += align_up(n * sizeof(T), 256);
    return r;
  }
};

#ifdef __HIPCC__
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Sum over the 32 lanes of each wave half (lanes 0-31 and 32-63 reduce independently).
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// Sum over the 32 lanes of each wave half with DPP adds (VALU only: no LDS crossbar round trips like the ds_bpermute
// behind __shfl_xor).  The total is valid in lanes 16..31 / 48..63 of the half only (use lane (lane & 31) == 31).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, true));
}
__device__ __forceinline__ float half_wave_sum_hi(float v) {
  v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);  // row_mirror
  v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  return v;
}

// v[lane] + v[lane ^ 32] in every lane, with the gfx950 VALU half swap (v_permlane32_swap) instead of a round trip
// through the LDS crossbar (ds_bpermute + s_waitcnt stalls the wave; this does not).  Inline asm: the builtin of
// this compiler returns the same register for both results.
__device__ __forceinline__ float lane_halves_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = half_wave_sum(v);
  v += __shfl_xor(v, 32);
  return v;
}

// Row of a 32x32 MFMA C/D tile held in accumulator register r of a lane in wave half `half`.
__device__ __forceinline__ constexpr int mfma32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
#endif

}  // namespace fil
