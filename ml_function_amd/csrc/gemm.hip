// N2  the dense layers of the zoo's MLPs (DnnLayer / HiddenLayer / Dense, core_layer/core_layer.py:102-118,201-226): fp32 GEMMs at
// batch-sized M and a few hundred columns -- y = x W (+ b, ReLU), dx = dz W^T, dW = x^T dz.  SURVEY 8 N2 leaves them to the library;
// at these shapes (4096 x 637 x 256 and smaller) hipBLASLt's fp32 kernels run at ~0.15 of the fp32 matrix rate (170 us of a 1.25 ms
// xDeepFM step for 5 GFLOP: profiles/r06_xdeepfm_model.txt), so they are hand-written too: one LDS-tiled v_mfma_f32_32x32x2_f32 kernel,
// exact fp32 chains, fixed summation order (split-K partials are summed in slice order by a second launch: deterministic).  Where it
// stands (tools/gemm_bench.py, replayed): the twelve GEMMs of the xDeepFM MLP incl. its logit head 151 us (180 before the two-waves-per-tile
// form and the row-contiguous LDS layout below; the library: ~190), the three 1.3-GFLOP ones 22-27 us = 0.3-0.4 of the matrix rate -- a 64 x 64
// tile stages 16 KB through registers into LDS per 64 MFMAs, and at 4096 x 256 there are not enough tiles for bigger ones.
//
//   C [M][N] = op(A) [M][K] . op(B) [K][N]  (+ bias[n], ReLU)
//   TA = 0: A is [M][K] row-major (k contiguous);  1: A is [K][M] (the transposed operand of dW = x^T dz)
//   TB = 0: B is [K][N] row-major (n contiguous);  1: B is [N][K] (the transposed operand of dx = dz W^T, W being [in][out])
// Workgroup = 4 waves = a 64 x 64 tile of C, wave = 32 x 32 (one MFMA tile), BK = 32 per LDS stage.  Both operands sit in LDS k-contiguous
// ([64][32 + 4 pad]): a lane (i, half) reads A[i][8t + 4 half .. + 3] as ONE 16-byte word = its operands of four MFMA steps (the
// reduction order inside a stage is permuted so that a wave half's k values are contiguous -- free in a GEMM, the same for A and B).
// Global -> registers two stages ahead, registers -> the other LDS buffer behind the MFMAs, one barrier per stage.
#include "common.h"
#include <cstdlib>

namespace fil {

typedef float gf32x4 __attribute__((ext_vector_type(4)));
constexpr int kGemmBM = 64, kGemmBK = 32, kGemmLd = kGemmBK + 4;

// one stage of an operand tile: 64 (rows of C's M or N side) x 32 (k) floats = two 16-byte pieces per thread, fetched by raw buffer
// loads (dword-aligned 16-byte loads are legal there -- the MLP's first layer has a row stride of 637 floats --, anything past the end of
// the operand reads as zero; what lies past the row or the k range INSIDE the operand is masked by hand).
//   contiguous along k (TR = false): piece u of thread t -> row (t >> 3) + 32 u, k quad t & 7;  one 16-byte LDS store each;
//   contiguous along the row index (TR = true): piece u -> k = (t >> 4) + 16 u, row quad t & 15: four rows of one k -> four 4-byte LDS stores.
template <bool TR, int ROWS, int T>   // ROWS of the tile (64 or 32), T threads of the workgroup
struct GemmStage {
  static constexpr int NP = ROWS * 8 / T;   // 16-byte pieces per thread
  gf32x4 v[NP];
  unsigned keep;   // bit 4 u + e: element e of piece u lies inside the operand's row / k range (applied at STORE time: a select at load
                   // time would make the thread wait for the load it has just issued)
  __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t& rs, long ld, int r0, int k0, int R, int K) {
    keep = 0;
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = threadIdx.x + u * T;
      if constexpr (!TR) {
        const int row = r0 + (idx >> 3), k = k0 + 4 * (idx & 7);
        v[u] = __builtin_bit_cast(gf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, row < R ? (int)(((long)row * ld + k) * 4) : 0x7ffffff0, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) keep |= (k + e < K ? 1u : 0u) << (4 * u + e);
      } else {
        const int k = k0 + idx / (ROWS / 4), row = r0 + 4 * (idx % (ROWS / 4));
        v[u] = __builtin_bit_cast(gf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, k < K ? (int)(((long)k * ld + row) * 4) : 0x7ffffff0, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) keep |= (row + e < R ? 1u : 0u) << (4 * u + e);
      }
    }
  }
  __device__ __forceinline__ void store(float* S) const {   // S: [ROWS][kGemmLd] (TR = false) or [32 k][ROWS + 4] (TR = true)
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int idx = threadIdx.x + u * T;
      gf32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (keep >> (4 * u + e)) & 1u ? v[u][e] : 0.f;
      if constexpr (!TR) *reinterpret_cast<gf32x4*>(S + (idx >> 3) * kGemmLd + 4 * (idx & 7)) = w;
      else *reinterpret_cast<gf32x4*>(S + (idx / (ROWS / 4)) * (ROWS + 4) + 4 * (idx % (ROWS / 4))) = w;   // [k][ROWS + 4]: one 16-byte store
      // (k-contiguous like the other operand, this piece was four 4-byte stores 144 floats apart: 4 banks for 16 lanes)
    }
  }
};

// EPI: 0 plain store, 1 + bias, 2 + bias then ReLU.  nsplit > 1: blockIdx.z takes the k range [z kchunk, (z + 1) kchunk) and writes
// its tile to part[z][M][N] (no epilogue: gemm_splitk_sum_kernel adds the slices in order).
// Pipeline: the operands of stage s+2 are requested while stage s computes (two register sets, the loop runs two stages per
// iteration) -- one stage of 16 MFMAs per wave is ~0.5 us, less than a round trip to L2; stage s+1 sits in the other LDS buffer.
// BN = 64: 4 waves, a 64 x 64 tile;  BN = 32: 2 waves, a 64 x 32 tile -- twice the workgroups, so that shapes with few tiles still put
// two or more INDEPENDENT workgroups on a CU (one's stores and barrier under the other's MFMAs).
// KW = 2: twice the waves -- waves w and w + T/128 share an output tile and take the two halves of every stage's k range (8 MFMAs each),
// the upper one's accumulators joining the lower one's through LDS after the loop (lower + upper: a fixed order).  With ONE wave per
// SIMD a stage was load -> LDS -> barrier -> LDS reads -> a chain of 16 dependent MFMAs with nothing beside it; two waves per SIMD
// cover each other's barriers and LDS round trips, and each thread stages half the pieces (the MLP's twelve GEMMs 180 -> 164 us).
template <bool TA, bool TB, int BN, int KW>
__global__ __launch_bounds__(4 * BN * KW) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                               const float* __restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
                                                               int epi, int kchunk, float* __restrict__ part) {
  constexpr int T = 4 * BN * KW;
  __shared__ __attribute__((aligned(16))) float As[2][kGemmBM * kGemmLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * kGemmLd];
  const int lane = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
  constexpr int NWT = BN == 64 ? 4 : 2;          // waves per output tile set
  const int wave = wave0 % NWT, kw = wave0 / NWT;   // kw: which half of a stage's k range (KW = 2)
  const int i = lane & 31, half = lane >> 5;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round robin, each XCD has its own L2 -- workgroup id of XCD id % 8 takes
  // tile (id % 8) (ntiles / 8) + id / 8 of the row-block-major list, so the column tiles of one block of A's rows (and their
  // neighbours) run on ONE XCD and that block comes out of HBM / the Infinity Cache once instead of once per column tile
  const int ntx = (N + BN - 1) / BN, nty = (M + kGemmBM - 1) / kGemmBM, ntile = ntx * nty;
  const int per = (ntile + 7) >> 3;
  const int tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile >= ntile) return;   // (whole workgroup, before any barrier)
  const int m0 = (tile / ntx) * kGemmBM, n0 = (tile % ntx) * BN;
  const int kb = blockIdx.z * kchunk, ke = min(K, kb + kchunk);
  const int wm = (BN == 64 ? (wave >> 1) : wave) * 32, wn = BN == 64 ? (wave & 1) * 32 : 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)std::min<long>((TA ? (long)K * lda : (long)M * lda) * 4, 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (int)std::min<long>((TB ? (long)N * ldb : (long)K * ldb) * 4, 0x7fffffffL), 0x00020000);
  GemmStage<TA, kGemmBM, T> sa[2];
  GemmStage<!TB, BN, T> sb[2];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nst = (ke - kb + kGemmBK - 1) / kGemmBK;
  auto fetch = [&](int s, int set) {   // (stages past the end: k >= ke everywhere -> out-of-range offsets, zeros)
    sa[set].load(ra, lda, m0, kb + s * kGemmBK, M, ke);
    sb[set].load(rb, ldb, n0, kb + s * kGemmBK, N, ke);
  };
  auto compute = [&](int cur) {
    // an operand staged k-contiguous is read as one 16-byte word per four steps; one staged row-contiguous ([k][rows + 4]) as four dwords
    const float* as = TA ? As[cur] + (4 * half) * (kGemmBM + 4) + wm + i : As[cur] + (wm + i) * kGemmLd + 4 * half;
    const float* bs = !TB ? Bs[cur] + (4 * half) * (BN + 4) + wn + i : Bs[cur] + (wn + i) * kGemmLd + 4 * half;
#pragma unroll
    for (int t0 = 0; t0 < 4 / KW; ++t0) {
      const int t = KW == 1 ? t0 : 2 * kw + t0;
      gf32x4 a4, b4;
      if constexpr (TA) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a4[e] = as[(8 * t + e) * (kGemmBM + 4)];
      } else {
        a4 = *reinterpret_cast<const gf32x4*>(as + 8 * t);
      }
      if constexpr (!TB) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = bs[(8 * t + e) * (BN + 4)];
      } else {
        b4 = *reinterpret_cast<const gf32x4*>(bs + 8 * t);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
    }
  };
  fetch(0, 0);
  fetch(1, 1);
  sa[0].store(As[0]);
  sb[0].store(Bs[0]);
  __syncthreads();
  for (int s = 0; s < nst; s += 2) {
    // stage s (LDS buffer 0, register set 0 is free again -> stage s+2); stage s+1's operands (set 1) go to buffer 1 behind the MFMAs
    fetch(s + 2, 0);
    compute(0);
    sa[1].store(As[1]);
    sb[1].store(Bs[1]);
    __syncthreads();
    if (s + 1 >= nst) break;
    fetch(s + 3, 1);
    compute(1);
    sa[0].store(As[0]);
    sb[0].store(Bs[0]);
    __syncthreads();
  }
  if constexpr (KW == 2) {   // the upper half's accumulators through LDS ([register][lane]: conflict-free; the stage buffers are done with)
    __syncthreads();
    float* xch = &As[0][0] + (wave * 16) * 64;
    if (kw == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kw == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += xch[r * 64 + lane];
  }
  // accumulator register r of lane (i, half): row (r & 3) + 8 (r >> 2) + 4 half, column i of the wave's 32 x 32 tile
  const int col = n0 + wn + i;
  float bv = 0.f;
  if (part == nullptr && epi >= 1 && col < N) bv = bias[col];
  float* dst = part != nullptr ? part + (long)blockIdx.z * M * N : C;
  const long ldd = part != nullptr ? (long)N : ldc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + mfma32_row(r, half);
    if (row < M && col < N) {
      float v = acc[r] + bv;
      if (part == nullptr && epi == 2) v = fmaxf(v, 0.f);
      dst[(long)row * ldd + col] = v;
    }
  }
}

// C[i] = sum over the nsplit slices of part[z][i], in slice order
__global__ __launch_bounds__(256) void gemm_splitk_sum_kernel(const float* __restrict__ part, float* __restrict__ C, long n, int nsplit, int N, long ldc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int z = 0; z < nsplit; ++z) t += part[(long)z * n + i];
  const long row = i / N;
  C[row * ldc + (i - row * N)] = t;
}

// 64 x 64 tiles (4 waves) where they give most CUs a workgroup, else 64 x 32 tiles (2 waves).  (Measured at 4096 x 256 x 637: 64 x 64 =
// 256 workgroups 27 us, 64 x 32 = 512 workgroups 37 us -- the kernel is bound by staging its operands (bytes through registers into LDS
// per MFMA), not by latency: the smaller tile re-reads A twice as often.)
static int gemm_bn(int M, int N) { return (long)cdiv(M, kGemmBM) * cdiv(N, 64) >= 192 ? 64 : 32; }
static int gemm_splits(int M, int N, int K) {
  const long tiles = (long)cdiv(M, kGemmBM) * cdiv(N, gemm_bn(M, N));
  if (tiles >= 192 || K < 512) return 1;                              // enough tiles to fill the chip, or nothing worth splitting
  const int want = (int)std::min<long>(16, std::max<long>(1, 768 / tiles));    // ~three 2-wave workgroups per CU
  const int kchunk = cdiv(cdiv(K, want), kGemmBK) * kGemmBK;
  return cdiv(K, kchunk);
}

}  // namespace fil

using namespace fil;

extern "C" size_t fil_gemm_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int ns = gemm_splits(M, N, K);
  return ns > 1 ? align_up((size_t)ns * M * N * sizeof(float), 256) : 0;
}

extern "C" int fil_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb, int ldc, int trans_a,
                            int trans_b, int epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  FIL_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && epilogue >= 0 && epilogue <= 2 && (trans_a == 0 || trans_a == 1) && (trans_b == 0 || trans_b == 1));
  if (M == 0 || N == 0) return FIL_OK;
  FIL_CHECK_ARG(C != nullptr && (K == 0 || (A != nullptr && B != nullptr)) && (epilogue == 0 || bias != nullptr));
  FIL_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N);
  if ((long)(trans_a ? K : M) * lda >= (1L << 29) || (long)(trans_b ? N : K) * ldb >= (1L << 29))
    return fail(FIL_ERR_UNSUPPORTED, "fil_gemm_f32: an operand of 2 GiB or more (32-bit buffer offsets)");
  hipStream_t st = (hipStream_t)stream;
  const int ns = epilogue != 0 ? 1 : gemm_splits(M, N, K);   // (a call with an epilogue is never split: its tiles finish in one workgroup)
  float* part = nullptr;
  if (ns > 1) {
    if (workspace == nullptr || workspace_bytes < fil_gemm_f32_workspace_bytes(M, N, K))
      return fail(FIL_ERR_WORKSPACE, "fil_gemm_f32: workspace %zu < %zu bytes", workspace_bytes, fil_gemm_f32_workspace_bytes(M, N, K));
    part = static_cast<float*>(workspace);
  }
  const int kchunk = ns > 1 ? cdiv(cdiv(K, ns), kGemmBK) * kGemmBK : std::max(K, 1);
  const int bn = gemm_bn(M, N);
  const dim3 grid(cdiv(cdiv(N, bn) * cdiv(M, kGemmBM), 8) * 8, 1, ns);   // (tiles in XCD-aware order: a multiple of 8 workgroups)
  ProfScope ps("gemm_f32", st, 2.0 * M * N * K);
  static const int kw_env = [] {   // FIL_GEMM_KW=1: one wave per output tile (A/B)
    const char* e = getenv("FIL_GEMM_KW");
    return e != nullptr && e[0] == '1' ? 1 : 2;
  }();
  const int kw = kw_env;
#define FIL_GEMM_L(TAV, TBV, BNV, KWV)                                                                                                  \
  hipLaunchKernelGGL((gemm_f32_kernel<TAV, TBV, BNV, KWV>), grid, dim3(4 * BNV * KWV), 0, st, A, B, C, bias, M, N, K, (long)lda, (long)ldb, \
                     (long)ldc, epilogue, kchunk, part)
#define FIL_GEMM(TAV, TBV)                                                    \
  do {                                                                        \
    if (bn == 64) {                                                           \
      if (kw == 2) FIL_GEMM_L(TAV, TBV, 64, 2); else FIL_GEMM_L(TAV, TBV, 64, 1); \
    } else {                                                                  \
      if (kw == 2) FIL_GEMM_L(TAV, TBV, 32, 2); else FIL_GEMM_L(TAV, TBV, 32, 1); \
    }                                                                         \
  } while (0)
  if (trans_a) {
    if (trans_b) FIL_GEMM(true, true); else FIL_GEMM(true, false);
  } else {
    if (trans_b) FIL_GEMM(false, true); else FIL_GEMM(false, false);
  }
#undef FIL_GEMM
#undef FIL_GEMM_L
  FIL_CHECK_LAUNCH();
  if (ns > 1) {
    const long n = (long)M * N;
    hipLaunchKernelGGL(gemm_splitk_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, C, n, ns, N, (long)ldc);
    FIL_CHECK_LAUNCH();
  }
  return FIL_OK;
}
