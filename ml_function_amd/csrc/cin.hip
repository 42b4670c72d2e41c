// A3  xDeepFM CIN: host-side drivers of the C ABI (fil_cin_*).  Device kernels: cin_kernels.h.
//
// Data flow (all internal tensors m-major, m = b*K + k; see cin_kernels.h):
//   fwd:  x [B,F,K] --transpose--> xT [M][F]  (kept in `saved` for the backward)
//         per layer l < L-1 (and the last one in mode 1): pack W_l -> Wf, cin_fwd3 -> x^l [M][HS_l] (saved) + pool partials
//         last layer (mode 0): wsum_L, cin_last_fwd -> pool;   head: pooled [B, L*K], out [B]
//   bwd:  head backward -> dP [B, L*K];  last layer (mode 0): rank-one dW_L, dbias_L, cin_last_bwd -> G^{L-1}, dX
//         per remaining layer: dbias (column sums of G), cin_dw3 (+ fixed-order reduce) -> dW_l,
//         pack W_l -> Wz, cin_dz3 -> G^{l-1}, dX;   finally dxT (+ Gx^0) --transpose--> dx [B,F,K]
#include "cin_kernels.h"
#include "cin_tail.h"
#include "cin_launch.h"
#include "cin_qtail.h"
#include "cin_qsplit.h"
#include "cin_qmerge.h"

#include <stdlib.h>

namespace fil {

struct CinShape {
  int B, F, K, L;
  int H[kCinMaxL];
  int Hp(int l) const { return l == 0 ? F : H[l - 1]; }
  int HS(int l) const { return 128 * cdiv(H[l], 128); }   // row stride of layer l's feature map / gradient
  int xps(int l) const { return l == 0 ? F : HS(l - 1); } // row stride of x^{l-1}
  long M() const { return (long)B * K; }
  int JT() const { return cin_jt_of(F); }
  int HSmax() const {
    int h = 0;
    for (int l = 0; l < L; ++l) h = std::max(h, HS(l));
    return h;
  }
};

static int check_shape(const char* fn, int B, int F, int K, int L, const int* H, CinShape& s) {
  if (B < 0 || F < 1 || K < 1 || L < 1 || H == nullptr) return fail(FIL_ERR_ARG, "%s: bad shape B=%d F=%d K=%d L=%d", fn, B, F, K, L);
  if (L > kCinMaxL) return fail(FIL_ERR_UNSUPPORTED, "%s: L=%d > %d", fn, L, kCinMaxL);
  if (F > kCinMaxFields) return fail(FIL_ERR_UNSUPPORTED, "%s: F=%d > %d fields", fn, F, kCinMaxFields);
  if ((long)B * K > (1L << 28)) return fail(FIL_ERR_UNSUPPORTED, "%s: B*K = %ld rows > 2^28 (row-split byte offsets of the dW kernel are 32-bit)", fn, (long)B * K);
  s.B = B; s.F = F; s.K = K; s.L = L;
  for (int l = 0; l < L; ++l) {
    if (H[l] < 1) return fail(FIL_ERR_ARG, "%s: H[%d]=%d", fn, l, H[l]);
    if (H[l] > kCinMaxH) return fail(FIL_ERR_UNSUPPORTED, "%s: H[%d]=%d > %d feature maps", fn, l, H[l], kCinMaxH);
    s.H[l] = H[l];
  }
  return FIL_OK;
}

static int chunks_of(int H) { return cdiv(H, 128); }
// rows per wave of the row-parallel kernels: 64 when that still yields about one wave per SIMD (1024 SIMDs)
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v != nullptr && *v != 0 ? atoi(v) : dflt;
}
// Process-level tuning knobs, read from the environment ONCE (first call into the library), never on the launch path:
//   FIL_CIN_MB=1|2        rows per wave (x32) of the row-parallel kernels (default: by M)
//   FIL_CIN_SYM=0         symmetric first-layer kernels off
//   FIL_CIN_DW_MB, FIL_CIN_DW_SPLITS, FIL_CIN_DZ_MB   launch shape of the dW / dZ kernels
//   FIL_CIN_TAIL_SPLITS   row splits of the fused tail's weight-gradient kernel
//   FIL_CIN_KSPLIT=0|4    reduction split of the row-parallel kernels over the 4 waves of a workgroup (default: by M)
//   FIL_CIN_QMERGE=0      quadratic tail: two weight-gradient launches (first layer, quadratic form) instead of the merged one
//   FIL_CIN_DZ2=0         ... its two data-gradient passes as two launches of the pair-symmetric dZ kernel instead of one two-pass launch
//   FIL_CIN_FWDQ=0        ... its forward as two 128-column launches + the pool kernel instead of the 256-column launch with fused pools
//   FIL_CIN_HEADFOLD=0    ... the pooled relayout + Dense(1) head as their own launch instead of the 256-column launch's epilogue
//   FIL_CIN_PACKFOLD=0    ... T's two operand layouts by a pack launch instead of by the T workgroups themselves (exact mode)
//   FIL_CIN_DWFOLD4=0     ... the merged weight-gradient launch folds PAIRS of row splits (4 tiles x 2 splits per workgroup) instead of quads
// Results are identical up to summation order whatever they say.  Per-call overrides for tests travel in `mode`
// (FIL_CIN_MB2, FIL_CIN_NOSYM), not through the environment.
struct Knobs {
  int mb, sym, dw_mb, dw_splits, dz_mb, tail_splits, tail_settle, tail_dz_mode, ksplit, dzs_mb, qtail, qmerge, dz2, fwdq, headfold, packfold, dwfold4;
};
static const Knobs& knobs() {
  static const Knobs k = {env_int("FIL_CIN_MB", 0), env_int("FIL_CIN_SYM", 1), env_int("FIL_CIN_DW_MB", 1), env_int("FIL_CIN_DW_SPLITS", 0),
                          env_int("FIL_CIN_DZ_MB", 1), env_int("FIL_CIN_TAIL_SPLITS", 0), env_int("FIL_CIN_TAIL_SETTLE", 0), env_int("FIL_CIN_TAIL_DZ_MODE", 0), env_int("FIL_CIN_KSPLIT", -1), env_int("FIL_CIN_DZS_MB", 0), env_int("FIL_CIN_QTAIL", 1), env_int("FIL_CIN_QMERGE", 1), env_int("FIL_CIN_DZ2", 1), env_int("FIL_CIN_FWDQ", 1), env_int("FIL_CIN_HEADFOLD", 1), env_int("FIL_CIN_PACKFOLD", 1), env_int("FIL_CIN_DWFOLD4", 1)};
  return k;
}
// per-call view of the knobs: the process defaults with the call's mode bits applied
struct CinTune {
  int mb_forced;
  bool sym, no_ksplit;
  explicit CinTune(int mode)
      : mb_forced((mode & FIL_CIN_MB2) ? 2 : knobs().mb), sym(knobs().sym != 0 && !(mode & FIL_CIN_NOSYM)), no_ksplit((mode & FIL_CIN_NOKSPLIT) != 0) {}
  int mb_rows(long M) const {
    if (mb_forced == 1 || mb_forced == 2) return mb_forced;
    return cdiv((int)std::min<long>(M, 1L << 30), 64) >= 768 ? 2 : 1;
  }
  // Small M (a strong-scaling shard: 512 samples x K = 16 is 256 blocks of 32 rows for 1024 SIMDs): a wave reduces over ALL
  // channels of its rows, so below one row block per SIMD the row-parallel kernels stop getting faster.  ks = 4 gives a row
  // block to the four waves of a workgroup, which split the reduction (h range / periods) and fold their partial sums through
  // LDS.  Used when that still leaves at most two waves per SIMD; exact kernels with 32-row blocks only.
  int ksplit(long M) const {
    if (knobs().ksplit == 0 || no_ksplit) return 1;
    if (mb_rows(M) != 1) return 1;
    if (knobs().ksplit == 4) return 4;
    return cdiv((int)std::min<long>(M, 1L << 30), 32) <= 512 ? 4 : 1;
  }
};

static const char* kFwdNames[kCinMaxL] = {"cin_fwd_l1", "cin_fwd_l2", "cin_fwd_l3", "cin_fwd_l4", "cin_fwd_l5", "cin_fwd_l6", "cin_fwd_l7", "cin_fwd_l8"};
static const char* kDwNames[kCinMaxL] = {"cin_bwd_dw_l1", "cin_bwd_dw_l2", "cin_bwd_dw_l3", "cin_bwd_dw_l4", "cin_bwd_dw_l5", "cin_bwd_dw_l6", "cin_bwd_dw_l7", "cin_bwd_dw_l8"};
static const char* kDzNames[kCinMaxL] = {"cin_bwd_dz_l1", "cin_bwd_dz_l2", "cin_bwd_dz_l3", "cin_bwd_dz_l4", "cin_bwd_dz_l5", "cin_bwd_dz_l6", "cin_bwd_dz_l7", "cin_bwd_dz_l8"};
// algorithmic flops of one layer GEMM: 2 * M * C * H
static double gemm_flops(long M, int Hp, int F, int H) { return 2.0 * (double)M * Hp * F * H; }

// ---- weight-gradient launch plan: waves over channel tiles x splits of the m range ~ one wave per SIMD
struct DwPlan {
  int MB, blocks_x, splits, rows_per_split, chunks;
};
static int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

// Row splits of the dW GEMM, from a launch-time model fitted on MI355X (profiles/r01_dw_split_sweep.txt).
// Workgroups are dealt evenly over the CUs, at most 3 resident per CU, in rounds of 3*CUs; a round with r resident
// workgroups per CU spends t(r) = {1.05, 1.39, 1.96} * 1e-4 ms per row of its split (two resident waves per SIMD
// already cover the MFMA pipe, so a third adds its full share of time); each split also costs one partial [C,H]
// write + re-read in the reduction (served from the Infinity Cache, about 6 TB/s).  Pick the cheapest split count.
static DwPlan dw_plan(long M, int C, int H) {
  DwPlan p;
  p.chunks = chunks_of(H);
  p.MB = knobs().dw_mb == 2 ? 2 : 1;   // 64 rows per wave measured slower (67 vs 122 TFLOP/s)
  const int waves_c = cdiv(C, 32 * p.MB);
  p.blocks_x = cdiv(waves_c, 4);
  const long tiles = (long)p.blocks_x * p.chunks;
  const long ncu = cu_count();
  const long unit = 2 * kDwDepth;
  auto rows_of = [&](int splits) { return std::max<long>(unit, ((M + splits - 1) / splits + unit - 1) / unit * unit); };
  int best = knobs().dw_splits;
  if (best <= 0) {
    static const double t_of[4] = {0.0, 1.05e-4, 1.39e-4, 1.96e-4};
    double best_ms = -1.0;
    // a split's byte offsets (rows * up to 1 KiB) must stay below 2^31: at most 2^20 rows per split
    for (int sp = (int)std::max<long>(1, (M + (1L << 20) - 1) >> 20); sp <= 256; ++sp) {
      const long rps = rows_of(sp);
      const long real = (M + rps - 1) / rps;           // splits that actually get rows
      if (real != sp && sp > 1) continue;              // same plan as a smaller sp
      long wgs = tiles * real;
      double per_row = 0.0;
      while (wgs > 0) {
        const long round = std::min(wgs, 3 * ncu);
        per_row += t_of[std::min<long>(3, (round + ncu - 1) / ncu)];
        wgs -= round;
      }
      const double ms = (double)rps * per_row * p.MB + (double)real * C * H * 4.0 / 6e9;
      if (best_ms < 0 || ms < best_ms) best = sp, best_ms = ms;
      if (rps == unit) break;
    }
  }
  best = std::min(std::max(best, 1), 256);
  const long rps = rows_of(best);
  p.rows_per_split = (int)rps;
  p.splits = (int)std::max<long>(1, (M + rps - 1) / rps);
  return p;
}

static int launch_dw3(hipStream_t st, const DwPlan& p, const float* gT, int HS, const float* xT, const float* xpT, int xps, float* part,
                      long M, int F, int Hp, int H, int symD = 0, int xtra = 0) {
  const int items = p.blocks_x * p.splits * p.chunks;
  const dim3 grid((items + 7) / 8 * 8);
#define FIL_DW3(MBV, ONES) \
  hipLaunchKernelGGL((cin_dw3_kernel<MBV, ONES>), grid, dim3(kCinThreads), 0, st, gT, HS, xT, xpT, xps, part, (int)M, F, Hp, H, p.rows_per_split, \
                     p.blocks_x, p.chunks, items, symD, xtra)
  if (xT == nullptr) {
    if (p.MB == 2) FIL_DW3(2, true); else FIL_DW3(1, true);
  } else {
    if (p.MB == 2) FIL_DW3(2, false); else FIL_DW3(1, false);
  }
#undef FIL_DW3
  return p.splits;
}

// ---- fused tail (cin_tail.h): geometry of the last two layers handled as one implicit GEMM with F+2 columns
struct TailGeom {
  bool on = false;
  int p = 0;                       // index of the lower tail layer (L-2); the upper one is L-1
  int NCB = 0, JP = 0, JT4 = 0, NQ = 0, JHp = 0;
  int Hpp = 0, Hq = 0, HL = 0, Cp = 0, C1 = 0;
  int periods = 0, tiles = 0;      // dZ stream (slot order of cin_pack_wz_kernel, one tile past the end)
  size_t uf_floats = 0, uz_floats = 0;
};
static TailGeom tail_geom(const CinShape& s) {   // what the tail WOULD look like (independent of mode: buffer sizes use it)
  TailGeom g;
  if (s.L < 3 || !cin_tail_supported(s.F)) return g;
  g.on = true;
  g.p = s.L - 2;
  g.NCB = cin_tail_ncb(s.F);
  g.JP = 16 * g.NCB;
  g.JT4 = cin_tail_jt4(s.F);
  g.NQ = cin_tail_nq(s.F);
  g.JHp = 4 * g.NQ;
  g.Hpp = s.H[g.p - 1];
  g.Hq = s.H[g.p];
  g.HL = s.H[s.L - 1];
  g.Cp = g.Hpp * s.F;
  g.C1 = g.Cp + 1;
  const int JT = s.JT();
  g.periods = cdiv(g.Hpp, cin_dz_h_per_period(JT));
  g.tiles = g.periods * cin_dz_tiles_per_period(JT) + 1;
  g.uf_floats = (size_t)g.Hpp * g.JT4 * 64 * g.NCB + (size_t)align_up(g.JP + 1, 64);   // Uf | consts (beff[JP], sum bias_L)
  g.uz_floats = (size_t)g.tiles * 64 * g.JHp;
  return g;
}
constexpr int kCinRetiredBits = 0;   // mode bits that no longer select anything: FIL_ERR_UNSUPPORTED (none at present)
// is the tail used by a call with these mode bits?
static bool tail_used(const CinShape& s, int mode) {
  if ((mode & (FIL_CIN_GENERAL | FIL_CIN_NOTAIL)) != 0) return false;
  const TailGeom g = tail_geom(s);
  if (!g.on) return false;
  return (mode & FIL_CIN_TAIL_ALWAYS) != 0 || 4 * g.JP <= 3 * g.Hq;
}
// Quadratic tail (cin_qtail.h): three layers, pair-symmetric first-layer kernels available, one 128-column chunk below the tail.
// The top two layers then cost F(F+1)/2 x H_1 products per row -- half of the fused tail's H_1 F (F+1), and no column padding.
// It takes ~10 more small launches than the fused tail: below ~16 K rows (a strong-scaling shard: the step is launch-latency bound
// there, B=512: 0.40 ms against 0.33) the fused tail stays the default; FIL_CIN_TAIL_ALWAYS lifts the size rule (tests).
static bool qtail_used(const CinShape& s, int mode, const CinTune& tune) {
  if (s.M() <= 16384 && (mode & FIL_CIN_TAIL_ALWAYS) == 0) return false;
  return tail_used(s, mode) && s.L == 3 && tune.sym && s.F >= 2 && s.F + 2 <= kQtConst && s.H[0] <= 128 && 3 * s.F + 1 <= s.HSmax() && (mode & FIL_CIN_NOQTAIL) == 0 &&
         knobs().qtail != 0;   // (3 F: its three [M][F] scratch arrays share one gradient buffer)
}
// ... with ONE weight-gradient and ONE data-gradient GEMM for the first layer and the quadratic form together (cin_qmerge.h)
static bool qmerge_used(const CinShape& s, int mode, const CinTune& tune) {
  return qtail_used(s, mode, tune) && 3 * s.F + 3 <= s.HSmax() && (mode & FIL_CIN_NOQMERGE) == 0 && knobs().qmerge != 0;   // (xe | gxR | dxR share one gradient buffer)
}
// ... on split-bf16 operands (cin_qsplit.h, FIL_CIN_BF16X3): where the merged kernels run in their full form and a split kernel exists
static bool qsplit_fwd_menu(int JT) { return JT >= 2 && JT <= 12 && JT % 2 == 0; }   // (F <= 41 on the merged tail: JT <= 12)
static bool qsplit_used(const CinShape& s, int mode, const CinTune& tune) {
  return (mode & FIL_CIN_BF16X3) != 0 && qmerge_used(s, mode, tune) && knobs().fwdq != 0 && knobs().dz2 != 0 && s.HS(0) == 128 &&
         qsplit_fwd_menu(cin_jt_sym(s.F));
}
static size_t qsplit_wb_bytes(const CinShape& s) { return (size_t)cin_qs_steps(s.F, cin_jt_sym(s.F)) * kQsStageBytes; }
static size_t qsplit_wzb_bytes(const CinShape& s) {     // one layer's slot-ordered weights as split planes (tiles x 8 steps x 3 KiB)
  const int jts = cin_jt_sym(s.F);
  return ((size_t)cdiv(s.F, cin_dz_h_per_period(jts)) * cin_dz_tiles_per_period(jts) + 1) * 8 * 3072;
}
static size_t qtail_wz_floats(const CinShape& s) {      // T in the dZ kernel's slot order
  const int jts = cin_jt_sym(s.F);
  return ((size_t)cdiv(s.F, cin_dz_h_per_period(jts)) * cin_dz_tiles_per_period(jts) + 1) * 32 * s.HS(0);
}
static size_t qtail_wsn_floats(const CinShape& s) { return (size_t)chunks_of(s.H[0]) * 2 * s.JT() * 128; }
static size_t qtail_saved_floats(const CinShape& s) {   // R | T | wsum_L | cvec | wsum_p | wsn_p | Wz(T), behind xT and the first layer's map
  return (size_t)s.M() * s.HS(0) + (size_t)s.F * s.F * s.H[0] + (size_t)s.H[1] * s.F + 128 + (size_t)s.H[0] * s.F + qtail_wsn_floats(s) +
         qtail_wz_floats(s);
}
struct TailDwPlan {
  int blocks_x, splits, rows_per_split;
};
static TailDwPlan tail_dw_plan(long M, int C1) {
  TailDwPlan p;
  p.blocks_x = cdiv(C1, 16 * kTailCbw);           // one workgroup per channel block and row split (its 4 waves quarter the split)
  const long unit = 4L * 4 * kTailDwDepth;        // a quarter of a split is a whole number of DEPTH-step groups
  // two workgroups per CU (two waves per SIMD: the second covers the first one's operand waits), all resident at once
  long want = knobs().tail_splits > 0 ? knobs().tail_splits : std::max<long>(1, 2L * cu_count() / p.blocks_x);
  want = std::max<long>(want, (M + (1L << 22) - 1) >> 22);   // byte offsets inside a split (rows * 256) stay below 2^31
  const long rows = std::max(unit, ((M + want - 1) / want + unit - 1) / unit * unit);
  p.rows_per_split = (int)rows;
  p.splits = (int)std::max<long>(1, (M + rows - 1) / rows);
  return p;
}

constexpr int kHeadChunk = 16;    // samples per block in the head partial reductions
constexpr int kColRows = 128;     // rows per block in the dbias (column-sum) partial reductions

static size_t saved_bytes(const CinShape& s) {
  size_t t = align_up((size_t)s.M() * s.F * sizeof(float), 256);  // xT
  for (int l = 0; l + 1 < s.L; ++l) t += align_up((size_t)s.M() * s.HS(l) * sizeof(float), 256);
  // fused tail: xT | maps 0..L-3 | Y [M][JP] | Uz | wsum_L   (whichever layout the call's mode picks must fit)
  const TailGeom g = tail_geom(s);
  if (g.on) {
    size_t u = align_up((size_t)s.M() * s.F * sizeof(float), 256);
    for (int l = 0; l < g.p; ++l) u += align_up((size_t)s.M() * s.HS(l) * sizeof(float), 256);
    u += align_up((size_t)s.M() * g.JP * sizeof(float), 256) + align_up((g.uz_floats + g.uf_floats) * sizeof(float), 256) +
         align_up((size_t)g.Hq * s.F * sizeof(float), 256);
    t = std::max(t, u);
  }
  if (g.on && s.L == 3) {   // quadratic tail: xT | map 0 | R | T | wsum_L | cvec
    const size_t u = align_up((size_t)s.M() * s.F * sizeof(float), 256) + align_up((size_t)s.M() * s.HS(0) * sizeof(float), 256) +
                     align_up(qtail_saved_floats(s) * sizeof(float), 256);
    t = std::max(t, u);
  }
  return t;
}
static size_t wf_floats(const CinShape& s) {
  size_t w = 0;
  for (int l = 0; l < s.L; ++l) {
    w = std::max(w, (size_t)chunks_of(s.H[l]) * s.Hp(l) * 2 * s.JT() * 128);
  }
  w = std::max(w, (size_t)chunks_of(s.H[0]) * s.F * 2 * cin_jt_sym(s.F) * 128);   // the pair-symmetric first layer
  w += (size_t)2 * 2 * s.JT() * 128;   // + the packed pooled weights of a fused last layer (<= 2 chunks)
  return w;
}
static int dz_periods(const CinShape& s, int l) { return cdiv(s.Hp(l), cin_dz_h_per_period(s.JT())); }
static size_t wz_floats(const CinShape& s) {
  const int jts = cin_jt_sym(s.F);   // symmetric first layer: its tile count and the pair-indexed dW sum both fit below
  size_t w = ((size_t)cdiv(s.F, cin_dz_h_per_period(jts)) * cin_dz_tiles_per_period(jts) + 1) * 32 * s.HS(0);
  w = std::max(w, (size_t)s.F * (s.F / 2 + 1) * s.H[0]);
  for (int l = 0; l < s.L; ++l) w = std::max(w, ((size_t)dz_periods(s, l) * cin_dz_tiles_per_period(s.JT()) + 1) * 32 * s.HS(l));
  return w;
}
// column chunks a layer's pooled partials may come in: its own, or (last layer pooled by the epilogue of the layer
// below) that layer's
static int pool_chunks(const CinShape& s, int l) { return std::max(chunks_of(s.H[l]), l > 0 ? chunks_of(s.H[l - 1]) : 1); }
static size_t fwd_ws_bytes(const CinShape& s) {
  size_t t = 0;
  for (int l = 0; l < s.L; ++l) t += align_up((size_t)pool_chunks(s, l) * s.M() * sizeof(float), 256);   // pool partials
  t += align_up((size_t)s.Hp(s.L - 1) * s.F * sizeof(float), 256);                                       // wsum of the last layer
  t += align_up(wf_floats(s) * sizeof(float), 256);                                                      // packed W
  t += align_up((size_t)kCinMaxH * sizeof(float), 256);                                                  // quadratic tail: zero bias of the R GEMM
  t += align_up(cin_x2_floats(s.M(), cin_x2_len(s.F)) * sizeof(float), 256);                             // wrapped rows of x (pair-symmetric forward)
  t += align_up((size_t)chunks_of(s.H[0]) * s.F * 2 * cin_jt_sym(s.F) * 128 * sizeof(float), 256);       // merged quadratic-tail forward: T's packed operand beside W1's
  t += align_up(qsplit_wb_bytes(s), 256);                                                                // ... its split-bf16 planes (FIL_CIN_BF16X3)
  return t;
}
// floats of the dW partial-sum buffer: the largest splits * C * H over the layers (both first-layer forms, so the
// size does not depend on the FIL_CIN_SYM knob) and the last layer's rank-one dW (C' = Hp, H' = F)
static size_t dw_part_floats(const CinShape& s) {
  size_t pmax = 0;
  for (int l = 0; l < s.L; ++l) pmax = std::max(pmax, (size_t)dw_plan(s.M(), s.Hp(l) * s.F, s.H[l]).splits * s.Hp(l) * s.F * s.H[l]);
  const int csym = s.F * (s.F / 2 + 1);
  pmax = std::max(pmax, (size_t)dw_plan(s.M(), csym, s.H[0]).splits * csym * s.H[0]);
  pmax = std::max(pmax, (size_t)dw_plan(s.M(), csym + s.F, s.H[0]).splits * (csym + s.F) * s.H[0]);   // quadratic tail: pairs + F single-field rows
  if (s.L == 3) pmax = std::max(pmax, (size_t)cin_dwq_plan(s.M(), csym + s.F, cu_count()).pairs * (csym + s.F) * 256);   // ... merged: 256 columns
  if (s.L == 3) pmax = std::max(pmax, (size_t)cin_dwqb_plan(s.M(), csym + s.F, cu_count()).splits * (csym + s.F) * 256);  // ... split-bf16: a partial per row split
  pmax = std::max(pmax, (size_t)dw_plan(s.M(), s.Hp(s.L - 1), s.F).splits * s.Hp(s.L - 1) * s.F);
  pmax = std::max(pmax, (size_t)dw_plan(s.M(), s.F, s.Hp(s.L - 1)).splits * s.Hp(s.L - 1) * s.F);   // (its swapped form)
  const TailGeom g = tail_geom(s);
  if (g.on) {   // fused tail: Q partials, then the dwsum_L partials of cin_tail_params_kernel
    pmax = std::max(pmax, (size_t)tail_dw_plan(s.M(), g.C1).splits * g.C1 * g.JP + (size_t)cdiv(g.Cp, kTailPc) * g.Hq * s.F);
  }
  return pmax;
}
static size_t bwd_ws_bytes(const CinShape& s) {
  const size_t LK = (size_t)s.L * s.K;
  const size_t M = (size_t)s.M();
  size_t t = 0;
  t += align_up((size_t)s.B * LK * sizeof(float), 256);                  // dP
  t += 2 * align_up(M * s.HSmax() * sizeof(float), 256);                 // G ping-pong (also the last layer's x*dP rows)
  t += align_up(dw_part_floats(s) * sizeof(float), 256);                 // dW partials
  const size_t nblk = (size_t)cdiv(std::max(1, s.B), kHeadChunk);
  const size_t ncol = (M + kColRows - 1) / kColRows;
  t += align_up(std::max(ncol * s.HSmax(), nblk * (LK + 1)) * sizeof(float), 256);   // colsum / head partials
  const size_t cl = (size_t)std::max(s.Hp(s.L - 1), s.L == 3 ? s.H[0] : 0) * s.F;   // (quadratic tail: the shortcut runs on layer L-2)
  t += 2 * align_up(cl * sizeof(float), 256);                            // wsum, v of the last layer
  t += align_up(wz_floats(s) * sizeof(float), 256);                      // packed W (slot order)
  t += 2 * align_up(M * s.F * sizeof(float), 256);                       // dxT, Gx^0
  t += align_up((size_t)s.F * s.F * kCinMaxH * sizeof(float), 256);      // quadratic tail: dT
  t += align_up(((M + 255) / 256 + 1) * kQtConst * sizeof(float), 256);  //                 column-sum partials of dP_L x, their sum
  t += align_up(((M + 255) / 256) * (LK + 1) * sizeof(float), 256);      //                 the dense head's block partials (merged launches)
  t += 2 * align_up(qsplit_wzb_bytes(s), 256);                           // FIL_CIN_BF16X3: W1s and Ts in slot order as split planes
  return t;
}

template <typename KernelT>
static void allow_lds(KernelT kernel, size_t sh) {
  if (sh > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
}

}  // namespace fil

using namespace fil;

extern "C" size_t fil_cin_saved_bytes(int B, int F, int K, int L, const int* H) {
  CinShape s;
  if (check_shape("fil_cin_saved_bytes", B, F, K, L, H, s) != FIL_OK || B == 0) return 0;
  return saved_bytes(s);
}

extern "C" size_t fil_cin_fwd_workspace_bytes(int B, int F, int K, int L, const int* H) {
  CinShape s;
  if (check_shape("fil_cin_fwd_workspace_bytes", B, F, K, L, H, s) != FIL_OK || B == 0) return 0;
  return fwd_ws_bytes(s);
}

extern "C" size_t fil_cin_bwd_workspace_bytes(int B, int F, int K, int L, const int* H) {
  CinShape s;
  if (check_shape("fil_cin_bwd_workspace_bytes", B, F, K, L, H, s) != FIL_OK || B == 0) return 0;
  return bwd_ws_bytes(s);
}

extern "C" int fil_cin_grad_ready_points(int B, int F, int K, int L, const int* H, int mode, int* point) {
  CinShape s;
  if (int rc = check_shape("fil_cin_grad_ready_points", B, F, K, L, H, s)) return rc;
  if (point == nullptr || mode < 0 || mode > 1023) return fail(FIL_ERR_ARG, "fil_cin_grad_ready_points: point == NULL or mode %d out of range", mode);
  if ((mode & kCinRetiredBits) != 0) return fail(FIL_ERR_UNSUPPORTED, "fil_cin_grad_ready_points: mode %d holds a retired bit (2: the split-bf16 experiment)", mode);
  if (B == 0) {                                         // empty batch: zero gradients, every slot at once
    for (int i = 0; i <= L; ++i) point[i] = 0;
    return 1;
  }
  int pt = 0, l = L - 1;
  point[L] = pt++;                                      // the dense head
  if (qmerge_used(s, mode, CinTune(mode))) {            // merged weight gradients: the first layer's come out first (the dense head's with them), then the top two layers'
    point[0] = point[L] = 0;
    pt = 1;
    point[1] = point[2] = pt++;
    return pt;
  }
  if (tail_used(s, mode)) {                             // fused tail: the two top layers' gradients come out of one group of launches
    point[L - 1] = point[L - 2] = pt++;
    l = L - 3;
  }
  for (; l >= 0; --l) point[l] = pt++;
  return pt;
}

extern "C" int fil_cin_fwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                           const float* dense_b, float* out, float* pooled, float* saved, int B, int F, int K, int L,
                           const int* H, int output_dim, int mode, void* workspace, size_t workspace_bytes, void* stream) {
  CinShape s;
  int rc = check_shape("fil_cin_fwd", B, F, K, L, H, s);
  if (rc != FIL_OK) return rc;
  if (mode < 0 || mode > 1023 || (mode & kCinRetiredBits) != 0)
    return fail(FIL_ERR_UNSUPPORTED, "fil_cin_fwd: mode %d (bits: 1 general kernels, 2 BF16X3, 4 MB2, 8 NOSYM, 16 X_TRANSPOSED, 32 NOTAIL, 64 TAIL_ALWAYS, 128 NOKSPLIT, 256 NOQTAIL, 512 NOQMERGE)", mode);
  const bool xt_in = (mode & FIL_CIN_X_TRANSPOSED) != 0;   // x is already [B*K][F] (fil_embed_gather_xt): no input transpose
  const CinTune tune(mode);
  const bool tail = tail_used(s, mode);                    // last two layers as one implicit GEMM (cin_tail.h)
  const bool qtail = qtail_used(s, mode, tune);            // ... as a quadratic form over field pairs (cin_qtail.h)
  const bool qmerge = qmerge_used(s, mode, tune);          // ... with merged launches (cin_qmerge.h)
  const bool qsplit = qsplit_used(s, mode, tune);          // ... on split-bf16 operands (cin_qsplit.h)
  const TailGeom tg = tail_geom(s);
  mode &= 1;
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(x && W && bias && pooled && saved);
  FIL_CHECK_ARG(output_dim != 1 || (dense_w && dense_b && out));
  if (workspace == nullptr || workspace_bytes < fwd_ws_bytes(s))
    return fail(FIL_ERR_WORKSPACE, "fil_cin_fwd: workspace %zu < %zu bytes", workspace_bytes, fwd_ws_bytes(s));
  hipStream_t st = (hipStream_t)stream;
  const long M = s.M();
  const int JT = s.JT(), MB = tune.mb_rows(M);
  Carver ws(workspace);
  PoolArgs pa;
  for (int l = 0; l < L; ++l) {
    pa.part[l] = ws.take<float>((size_t)pool_chunks(s, l) * M);
    pa.chunks[l] = chunks_of(H[l]);
  }
  float* wsum = ws.take<float>((size_t)s.Hp(L - 1) * F);
  float* Wf = ws.take<float>(wf_floats(s));
  float* qt_zbias = ws.take<float>((size_t)kCinMaxH);
  const int XL = cin_x2_len(F);
  float* x2T = ws.take<float>(cin_x2_floats(M, XL));
  const bool need_x2 = tune.sym;   // the pair-symmetric forward kernel reads the wrapped rows
  float* WfT = ws.take<float>((size_t)chunks_of(H[0]) * F * 2 * cin_jt_sym(F) * 128);
  u32x4* Wb = reinterpret_cast<u32x4*>(ws.take<unsigned char>(qsplit_wb_bytes(s)));
  Carver sv(saved);
  float* xT_own = sv.take<float>((size_t)M * F);       // (unused when x arrives transposed; the layout of `saved` stays the same)
  const float* xT = xt_in ? x : xT_own;
  // fused tail: its slices of `saved` (behind xT and the maps of the layers below it)
  float *tailY = nullptr, *tailUz = nullptr, *tailBmT = nullptr;
  float *qtR = nullptr, *qtT = nullptr, *qtWsumL = nullptr, *qtCvec = nullptr, *qtWsumP = nullptr, *qtWsnP = nullptr, *qtWzT = nullptr;
  if (qtail) {
    Carver pv(saved);
    (void)pv.take<float>((size_t)M * F);
    (void)pv.take<float>((size_t)M * s.HS(0));
    float* q = pv.take<float>(qtail_saved_floats(s));
    qtR = q;
    qtT = qtR + (size_t)M * s.HS(0);
    qtWsumL = qtT + (size_t)F * F * H[0];
    qtCvec = qtWsumL + (size_t)H[1] * F;
    qtWsumP = qtCvec + 128;
    qtWsnP = qtWsumP + (size_t)H[0] * F;
    qtWzT = qtWsnP + qtail_wsn_floats(s);
  } else if (tail) {
    Carver pv(saved);
    (void)pv.take<float>((size_t)M * F);
    for (int l = 0; l < tg.p; ++l) (void)pv.take<float>((size_t)M * s.HS(l));
    tailY = pv.take<float>((size_t)M * tg.JP);
    tailUz = pv.take<float>(tg.uz_floats + tg.uf_floats);   // Uz | Uf | consts: one buffer, one clear
    tailBmT = pv.take<float>((size_t)tg.Hq * F);
  }
  // the exact pair-symmetric first layer + fused tail (the north-star path): every preparation job that depends on the inputs
  // alone -- x transpose, first-layer weight pack, pooled weights of the last layer, clearing the tail's operand buffers -- in ONE
  // launch instead of four
  const bool prep_fused = tail && tune.sym;
  if (qtail) {
    FIL_CHECK_ARG(W[0] && W[L - 1] && W[L - 2]);
    ProfScope ps("cin_fwd_prep", st, 2.0 * M * F * sizeof(float));
    const int JTs = cin_jt_sym(F), chunks0 = chunks_of(H[0]);
    const long npack = (long)chunks0 * F * 2 * JTs * 128;
    // (merged forward: the x transposes ride in the NEXT launch, behind the T workgroups, whose latency chain is the longer one -- this
    // launch is the weight work alone.  Measured the other way round, transposes here and T alone there: 9.8 + 11.4 us against
    // 4.2 + 12.5)
    const bool fq = qmerge && knobs().fwdq != 0 && s.HS(0) == 128;
    const int nt = fq ? 0 : B, npk = (int)std::min<long>((npack + 255) / 256, 1024), nwl = cdiv(tg.Hq * F, 8), nwp = cdiv(tg.Hpp * F, 8);
    const size_t sh = (xt_in || fq) ? 0 : (size_t)F * (K + 1) * sizeof(float);
    allow_lds(cin_qtail_prep_kernel, sh);
    hipLaunchKernelGGL(cin_qtail_prep_kernel, dim3(nt + npk + nwl + nwp), dim3(256), sh, st, x, xT_own, F, K, nt, W[0], Wf, H[0], 2 * JTs, chunks0, npk,
                       W[L - 1], qtWsumL, tg.Hq, tg.HL, nwl, W[L - 2], qtWsumP, qtWsnP, tg.Hpp, 2 * JT, chunks_of(tg.Hpp), x2T, XL, xt_in ? 1 : 0);
  } else if (prep_fused) {
    FIL_CHECK_ARG(W[0] && W[L - 1]);
    ProfScope ps("cin_fwd_prep", st, 2.0 * M * F * sizeof(float));
    const int JTs = cin_jt_sym(F), chunks0 = chunks_of(H[0]);
    const long npack = (long)chunks0 * F * 2 * JTs * 128;
    const int nt = B, npk = (int)std::min<long>((npack + 255) / 256, 1024), nws = cdiv(tg.Hq * F, 8), nz = 64;
    const size_t sh = xt_in ? 0 : (size_t)F * (K + 1) * sizeof(float);
    allow_lds(cin_fwd_prep_kernel, sh);
    hipLaunchKernelGGL(cin_fwd_prep_kernel, dim3(nt + npk + nws + nz), dim3(256), sh, st, x, xT_own, F, K, nt, W[0], Wf, H[0], 2 * JTs, chunks0,
                       npk, W[L - 1], tailBmT, tg.Hq, tg.HL, nws, reinterpret_cast<float4*>(tailUz), (long)((tg.uz_floats + tg.uf_floats) / 4), x2T, XL,
                       xt_in ? 1 : 0);
  } else if (!xt_in || need_x2) {
    ProfScope ps("cin_transpose_in", st, 2.0 * M * F * sizeof(float));
    hipLaunchKernelGGL(cin_transpose_in_kernel, dim3(B), dim3(256), xt_in ? 0 : (size_t)F * (K + 1) * sizeof(float), st, x, xT_own, F, K,
                       need_x2 ? x2T : nullptr, XL, xt_in ? 1 : 0);
  }
  FIL_CHECK_LAUNCH();
  const float* xpT = xT;
  bool fused_last = false;   // the last layer's sum-pool was produced by the epilogue of the layer below
  bool head_done = false;    // merged forward: pooled and out came out of cin_fwdq_kernel's epilogue
  for (int l = 0; l < L; ++l) {
    FIL_CHECK_ARG(W[l] && bias[l]);
    if (qtail && qmerge && l == 0 && knobs().fwdq != 0 && s.HS(0) == 128) {
      // ---- merged quadratic tail, forward (cin_qmerge.h): [x1 | R] = pairs(x) [W1s | Ts] in ONE launch of 256 columns, all three
      // sum-pools in its epilogue.  T (and its packed operand copies) depend on the weights alone: they come first.
      const int p = tg.p, lL = L - 1, Hpp = tg.Hpp, Hq = tg.Hq, HS0 = s.HS(0);
      FIL_CHECK_ARG(W[p] && W[lL] && bias[p] && bias[lL]);
      const int JTs = cin_jt_sym(F), chunks = chunks_of(Hpp);
      float* x1T = sv.take<float>((size_t)M * HS0);   // (the first layer's map: same place in `saved` as on the other paths)
      {
        ProfScope ps("cin_tail_prep", st);
        // T / cvec (weights only) and, beside them, the x -> xT / wrapped-row transposes (x only)
        // (x as given and K a power of two <= 64: the transposes go by 64-row blocks of whole samples, cin_transpose_block_body)
        int ks = -1;
        if (!xt_in && K <= 64 && (K & (K - 1)) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
          for (ks = 0; (1 << ks) < K;) ++ks;
        const size_t sh_x = ks >= 0 ? (size_t)(64 >> ks) * F * (K + 1) : (xt_in ? (size_t)64 * (F | 1) : (size_t)F * (K + 1));
        const int tiles = cdiv(F, cin_dz_h_per_period(JTs)) * cin_dz_tiles_per_period(JTs) + 1;
        // exact mode: every T workgroup writes its column of both operand layouts itself (one workgroup per column) -- no pack launch;
        // split-bf16 mode: the planes of [W1s | Ts] need whole rows of T, the pack launch stays
        const bool fold = !qsplit && knobs().packfold != 0;
        const QtPackFold pf = fold ? QtPackFold{WfT, qtWzT, JTs, chunks, HS0, tiles} : QtPackFold{nullptr, nullptr, 0, 0, 0, 0};
        const size_t sh = std::max(cin_qtail_t_lds_floats(F, Hq, fold), sh_x) * sizeof(float);
        allow_lds(cin_qtail_t_x_kernel, sh);
        const int nx = (ks >= 0 || xt_in) ? (int)((M + 63) / 64) : B, nT = cin_qtail_t_wgs(Hpp);
        hipLaunchKernelGGL(cin_qtail_t_x_kernel, dim3(nT + nx), dim3(256), sh, st, W[p], qtWsumL, bias[p], bias[lL], tg.HL, qtT, qtCvec, qt_zbias, Hpp, F, Hq,
                           nT, x, xT_own, K, x2T, XL, xt_in ? 1 : 0, ks, (ks >= 0 || xt_in) ? (long)M : 0L, pf);
        if (!fold) {
          const long npack = (long)chunks * F * 2 * JTs * 128;
          const int nbf = (int)std::min<long>((npack + 255) / 256, 1024), nbz = (int)std::min<long>(((long)tiles * 32 * HS0 + 255) / 256, 1024);
          // (split-bf16 mode: + the forward's planes of [W1s | Ts], from W1 and T themselves)
          const int NTq = qsplit ? cin_qs_steps(F, JTs) : 0, nbq = qsplit ? std::min(cdiv(NTq * 512, 256), 512) : 0;
          hipLaunchKernelGGL(cin_qtail_pack_kernel, dim3(nbf + nbz + nbq), dim3(256), 0, st, qtT, WfT, qtWzT, F, Hpp, JTs, chunks, nbf, HS0, tiles, nbz, W[0],
                             H[0], Wb, NTq);
        }
      }
      FIL_CHECK_LAUNCH();
      {
        const double algo = gemm_flops(M, F, F, H[0]) + gemm_flops(M, Hpp, F, Hq) + gemm_flops(M, Hq, F, tg.HL);   // all three layers of the reference graph
        ProfScope ps("cin_fwd_q", st, algo, gemm_flops(M, 1, F * (F / 2 + 1), H[0]) + gemm_flops(M, 1, F * (F / 2 + 1), Hpp));
        // (K a power of two <= 32: a wave's rows are whole samples, and the pooled relayout + Dense(1) head ride in the epilogue)
        CinHeadFold hf{};
        if (K <= 32 && (K & (K - 1)) == 0 && knobs().headfold != 0) {
          int ks = 0;
          while ((1 << ks) < K) ++ks;
          hf = CinHeadFold{pooled, output_dim == 1 ? out : nullptr, dense_w, dense_b, ks, (int)(L * K), p * K, lL * K};
          head_done = true;
        }
        if (qsplit) {
          // split-bf16 operands (their planes came out of the pack launch above): the same GEMM on the bf16 pipe
          const int NT = cin_qs_steps(F, JTs);
          if (!cin_launch_fwdq_b(st, JTs, x2T, XL, Wb, NT, bias[0], qtWsnP, JT, qtCvec, x1T, qtR, HS0, const_cast<float*>(pa.part[0]),
                                 const_cast<float*>(pa.part[p]), const_cast<float*>(pa.part[lL]), (int)M, F, H[0], hf))
            return fail(FIL_ERR_UNSUPPORTED, "fil_cin_fwd: no split-bf16 forward kernel for JT=%d (F=%d)", JTs, F);
        } else if (!cin_launch_fwdq(st, JTs, x2T, XL, Wf, WfT, bias[0], qtWsnP, JT, qtCvec, x1T, qtR, HS0, const_cast<float*>(pa.part[0]),
                                    const_cast<float*>(pa.part[p]), const_cast<float*>(pa.part[lL]), (int)M, F, H[0], hf))
          return fail(FIL_ERR_UNSUPPORTED, "fil_cin_fwd: no merged forward kernel for JT=%d (F=%d)", JTs, F);
        pa.chunks[0] = pa.chunks[p] = pa.chunks[lL] = 1;
      }
      FIL_CHECK_LAUNCH();
      break;
    }
    if (qtail && l == tg.p) {
      // ---- quadratic tail (cin_qtail.h): R = (pairs of x) T through the first layer's pair-symmetric forward kernel,
      // pool_L = <x1, R> + <x, c> + const, pool_p through the pooled-weights shortcut
      const int lL = L - 1, Hpp = tg.Hpp, Hq = tg.Hq, HS0 = s.HS(0);
      FIL_CHECK_ARG(W[lL] && bias[lL]);
      const int JTs = cin_jt_sym(F), chunks = chunks_of(Hpp);
      {
        ProfScope ps("cin_tail_prep", st);
        // (wsum_L, wsum_p and its MFMA operand copy came out of the preparation launch)
        const size_t sh = cin_qtail_t_lds_floats(F, Hq, false) * sizeof(float);
        allow_lds(cin_qtail_t_kernel, sh);
        hipLaunchKernelGGL(cin_qtail_t_kernel, dim3(cin_qtail_t_wgs(Hpp)), dim3(256), sh, st, W[l], qtWsumL, bias[l], bias[lL], tg.HL, qtT, qtCvec, qt_zbias, Hpp, F, Hq);
        // T in the forward kernel's operand layout (workspace) and in the dZ kernel's slot order (saved for the backward): one launch
        const long npack = (long)chunks * F * 2 * JTs * 128;
        const int tiles = cdiv(F, cin_dz_h_per_period(JTs)) * cin_dz_tiles_per_period(JTs) + 1;
        const int nbf = (int)std::min<long>((npack + 255) / 256, 1024), nbz = (int)std::min<long>(((long)tiles * 32 * HS0 + 255) / 256, 1024);
        hipLaunchKernelGGL(cin_qtail_pack_kernel, dim3(nbf + nbz), dim3(256), 0, st, qtT, Wf, qtWzT, F, Hpp, JTs, chunks, nbf, HS0, tiles);
      }
      FIL_CHECK_LAUNCH();
      {
        const double algo = gemm_flops(M, Hpp, F, Hq) + gemm_flops(M, Hq, F, tg.HL);   // the two layers of the reference graph
        ProfScope ps("cin_fwd_tail", st, algo, gemm_flops(M, 1, F * (F / 2 + 1), Hpp));
        const int ks = tune.ksplit(M);
        // (the kernel's own sum-pool output is not used: it goes to the last layer's slot, which the pool kernel below overwrites)
        cin_launch_fwd3_sym(st, MB, JTs, dim3(ks == 4 ? cdiv((int)M, 32) : cdiv((int)M, 128 * MB), chunks), xT, x2T, XL, Wf, qt_zbias, qtR, HS0,
                            const_cast<float*>(pa.part[lL]), (int)M, F, Hpp, ks);
      }
      FIL_CHECK_LAUNCH();
      {
        ProfScope ps("cin_tail_pool", st);
        const float* wsn = qtWsnP;
        float* pp = const_cast<float*>(pa.part[l]);
        float* pL = const_cast<float*>(pa.part[lL]);
        const dim3 grid((int)((M + 127) / 128));
#define FIL_QP(JTV) \
  case JTV: hipLaunchKernelGGL((cin_qtail_pool2_kernel<JTV>), grid, dim3(256), 0, st, xT, xpT, s.xps(l), wsn, qtR, HS0, qtCvec, pp, pL, (int)M, F, Hpp); break;
        switch (JT) { FIL_QP(4) FIL_QP(8) FIL_QP(12) FIL_QP(16) FIL_QP(20) FIL_QP(24) FIL_QP(28) FIL_QP(32) }
#undef FIL_QP
        pa.chunks[l] = pa.chunks[lL] = 1;
      }
      FIL_CHECK_LAUNCH();
      break;
    }
    if (tail && l == tg.p) {
      // ---- fused tail: layers p = L-2 and L-1 through Ueff = W_p [1 | wsum_L]: F+1 output columns instead of H_p
      const int lL = L - 1;
      FIL_CHECK_ARG(W[lL] && bias[lL]);
      float* Y = tailY;
      float* Uz = tailUz;
      float* bmT = tailBmT;
      float* Uf = Uz + tg.uz_floats;
      float* consts = Uf + (size_t)tg.Hpp * tg.JT4 * 64 * tg.NCB;
      {
        ProfScope ps("cin_tail_prep", st);
        if (!prep_fused) {
          hipLaunchKernelGGL(cin_tail_wsum_kernel, dim3(cdiv(tg.Hq * F, 8)), dim3(256), 0, st, W[lL], bmT, tg.Hq, F, tg.HL);
          // (padding of the operand layouts -- f >= F, j > F, spare slots -- must be zero)
          (void)hipMemsetAsync(Uz, 0, (tg.uz_floats + tg.uf_floats) * sizeof(float), st);
        }
        const size_t sh = (size_t)(F + 1) * (((tg.Hq + 3) & ~3) + 4) * sizeof(float);
        allow_lds(cin_tail_ueff_kernel, sh);
        hipLaunchKernelGGL(cin_tail_ueff_kernel, dim3(cdiv(tg.C1, kTailUc)), dim3(256), sh, st, W[l], bias[l], bmT, bias[lL], tg.HL, Uf, Uz, consts,
                           tg.Hpp, F, tg.Hq, tg.JT4, tg.JP, JT, tg.JHp);
      }
      FIL_CHECK_LAUNCH();
      {
        const double algo = gemm_flops(M, tg.Hpp, F, tg.Hq) + gemm_flops(M, tg.Hq, F, tg.HL);   // the two layers of the reference graph
        ProfScope ps("cin_fwd_tail", st, algo, 2.0 * (double)M * tg.Cp * (F + 1));
        const int RB = tune.mb_rows(M) == 2 ? 4 : 2;
        TailFwdArgs a{xT, xpT, s.xps(l), Uf, consts, Y, tg.JP, const_cast<float*>(pa.part[l]), const_cast<float*>(pa.part[lL]), (int)M, F, tg.Hpp,
                      tune.ksplit(M)};
        cin_launch_tail_fwd(st, RB, tg.JT4, tg.NCB, a);
        pa.chunks[l] = pa.chunks[lL] = 1;
      }
      FIL_CHECK_LAUNCH();
      break;
    }
    const int Hp = s.Hp(l), Hl = H[l], xps = s.xps(l);
    float* xoutT = l + 1 < L ? sv.take<float>((size_t)M * s.HS(l)) : nullptr;
    float* part = const_cast<float*>(pa.part[l]);
    // mode 0, last layer: only its sum-pool is observable -> contract with wsum[c] = sum_n W[c,n].  When the layer
    // below it runs the general (non pair-symmetric) forward kernel, that kernel's epilogue does it (fused_last).
    const bool fuse_next = mode == 0 && l == L - 2 && !(l == 0 && tune.sym);
    if (l == L - 1 && mode == 0) {
      if (!fused_last) {
        const size_t sh = (size_t)Hp * ((F + 3) & ~3) * sizeof(float);
        ProfScope ps("cin_last_fwd", st, 2.0 * (double)M * Hp * F);
        hipLaunchKernelGGL(cin_wsum_kernel, dim3(cdiv(Hp * F, 8)), dim3(256), 0, st, W[l], wsum, Hp * F, Hl);
        allow_lds(cin_last_fwd_kernel, sh);
        hipLaunchKernelGGL(cin_last_fwd_kernel, dim3(cdiv((int)M, kLastRows)), dim3(256), sh, st, xT, xpT, xps, wsum, bias[l], part, (int)M, F, Hp, Hl);
        pa.chunks[l] = 1;
      }
    } else {
      const int chunks = chunks_of(Hl);
      if (l == 0 && tune.sym) {
        // first layer: x^{l-1} = x, reduce over unordered field pairs (half the steps)
        const int JTs = cin_jt_sym(F);
        const long npack = (long)chunks * F * 2 * JTs * 128;
        if (!prep_fused)
          hipLaunchKernelGGL(cin_pack_wf_sym_kernel, dim3((int)std::min<long>((npack + 255) / 256, 2048)), dim3(256), 0, st, W[l], Wf, F, Hl, 2 * JTs, chunks);
        ProfScope ps(kFwdNames[l], st, gemm_flops(M, Hp, F, Hl), gemm_flops(M, 1, F * (F / 2 + 1), Hl));   // (executed: unordered pairs)
        const int ks = tune.ksplit(M);
        cin_launch_fwd3_sym(st, MB, JTs, dim3(ks == 4 ? cdiv((int)M, 32) : cdiv((int)M, 128 * MB), chunks), xT, x2T, XL, Wf, bias[l], xoutT, s.HS(l), part,
                            (int)M, F, Hl, ks);
      } else {
        const long npack = (long)chunks * Hp * 2 * JT * 128;
        hipLaunchKernelGGL(cin_pack_wf_kernel, dim3((int)std::min<long>((npack + 255) / 256, 2048)), dim3(256), 0, st, W[l], Wf, Hp, F, Hl, 2 * JT, chunks);
        const float* wsn = nullptr;
        if (fuse_next) {
          FIL_CHECK_ARG(W[l + 1] && bias[l + 1]);
          float* wsn_buf = Wf + (size_t)npack;   // behind this layer's packed weights
          hipLaunchKernelGGL(cin_wsum_wsn_kernel, dim3(cdiv(Hl * F, 8)), dim3(256), 0, st, W[l + 1], wsum, Hl * F, H[l + 1], wsn_buf, Hl, F, 2 * JT,
                             chunks);
          wsn = wsn_buf;
          pa.chunks[l + 1] = chunks;
          fused_last = true;
        }
        ProfScope ps(kFwdNames[l], st, gemm_flops(M, Hp, F, Hl) + (fuse_next ? 2.0 * (double)M * Hl * F : 0.0));
        cin_launch_fwd3(st, MB, JT, dim3(cdiv((int)M, 128 * MB), chunks), xT, xpT, xps, Wf, bias[l], xoutT, s.HS(l), part, (int)M, F, Hp, Hl,
                        wsn, fuse_next ? bias[l + 1] : nullptr, fuse_next ? H[l + 1] : 0, fuse_next ? const_cast<float*>(pa.part[l + 1]) : nullptr);
      }
    }
    FIL_CHECK_LAUNCH();
    xpT = xoutT;
  }
  if (!head_done) {
    ProfScope ps("cin_head_fwd", st);
    hipLaunchKernelGGL(cin_head_fwd_kernel, dim3(cdiv(B, kHeadSamples)), dim3(256), (size_t)kHeadSamples * L * K * sizeof(float), st, pa, dense_w, dense_b, pooled,
                       output_dim == 1 ? out : nullptr, B, K, L);
  }
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_cin_bwd(const float* x, const float* const* W, const float* const* bias, const float* dense_w,
                           const float* pooled, const float* saved, const float* g, float* dx, float* const* dW,
                           float* const* dbias, float* ddense_w, float* ddense_b, int B, int F, int K, int L, const int* H,
                           int output_dim, int mode, void* const* grad_ready_events, void* workspace, size_t workspace_bytes,
                           void* stream) {
  CinShape s;
  int rc = check_shape("fil_cin_bwd", B, F, K, L, H, s);
  if (rc != FIL_OK) return rc;
  if (mode < 0 || mode > 1023 || (mode & kCinRetiredBits) != 0)
    return fail(FIL_ERR_UNSUPPORTED, "fil_cin_bwd: mode %d (bits: 1 general kernels, 2 BF16X3, 4 MB2, 8 NOSYM, 16 X_TRANSPOSED, 32 NOTAIL, 64 TAIL_ALWAYS, 128 NOKSPLIT, 256 NOQTAIL, 512 NOQMERGE)", mode);
  const bool xt_in = (mode & FIL_CIN_X_TRANSPOSED) != 0;
  const CinTune tune(mode);
  const bool tail = tail_used(s, mode);
  const bool qtail = qtail_used(s, mode, tune);
  const bool qmerge = qmerge_used(s, mode, tune);
  const bool qsplit = qsplit_used(s, mode, tune);
  const TailGeom tg = tail_geom(s);
  mode &= 1;
  FIL_CHECK_ARG(W && dW && dbias);
  hipStream_t st = (hipStream_t)stream;
  const size_t LK = (size_t)L * K;
  // grad_ready_events[l] (l < L): recorded once dW[l] and dbias[l] are final; [L]: the dense head's gradients.  The
  // data-parallel caller makes a side stream wait on them and starts each layer's all-reduce while the rest of the
  // backward is still running (gradients become final from the top layer down).
  auto ready = [&](int slot) {
    if (grad_ready_events != nullptr && grad_ready_events[slot] != nullptr) (void)hipEventRecord((hipEvent_t)grad_ready_events[slot], st);
  };
  if (B == 0) {  // empty batch: parameter gradients are zero
    for (int l = 0; l < L; ++l) {
      (void)hipMemsetAsync(dW[l], 0, (size_t)s.Hp(l) * F * H[l] * sizeof(float), st);
      (void)hipMemsetAsync(dbias[l], 0, (size_t)H[l] * sizeof(float), st);
    }
    if (output_dim == 1) {
      (void)hipMemsetAsync(ddense_w, 0, LK * sizeof(float), st);
      (void)hipMemsetAsync(ddense_b, 0, sizeof(float), st);
    }
    for (int l = 0; l <= L; ++l) ready(l);
    return FIL_OK;
  }
  FIL_CHECK_ARG(g && dx && saved && (x || !xt_in));
  FIL_CHECK_ARG(output_dim != 1 || (dense_w && pooled && ddense_w && ddense_b));
  if (workspace == nullptr || workspace_bytes < bwd_ws_bytes(s))
    return fail(FIL_ERR_WORKSPACE, "fil_cin_bwd: workspace %zu < %zu bytes", workspace_bytes, bwd_ws_bytes(s));
  const long M = s.M();
  const int JT = s.JT();
  Carver ws(workspace);
  float* dP = ws.take<float>((size_t)B * LK);
  float* Gbuf[2];
  Gbuf[0] = ws.take<float>((size_t)M * s.HSmax());
  Gbuf[1] = ws.take<float>((size_t)M * s.HSmax());
  float* part = ws.take<float>(dw_part_floats(s));
  const int nblk = cdiv(B, kHeadChunk);
  const int ncol = (int)((M + kColRows - 1) / kColRows);
  float* small = ws.take<float>(std::max((size_t)ncol * s.HSmax(), (size_t)nblk * (LK + 1)));
  const size_t cl = (size_t)s.Hp(L - 1) * F;
  const size_t cl_buf = (size_t)std::max(s.Hp(L - 1), L == 3 ? H[0] : 0) * F;
  float* wsum = ws.take<float>(cl_buf);
  float* vlast = ws.take<float>(cl_buf);
  float* Wz = ws.take<float>(wz_floats(s));
  float* dxT = ws.take<float>((size_t)M * F);
  float* gx0T = ws.take<float>((size_t)M * F);
  float* qt_dT = ws.take<float>((size_t)F * F * kCinMaxH);
  const int qt_ndc = (int)((M + 255) / 256);
  float* qt_dcpart = ws.take<float>((size_t)(qt_ndc + 1) * kQtConst);   // block partials | their sum
  float* qt_hpart = ws.take<float>((size_t)qt_ndc * (LK + 1));           // merged launches: the dense head's block partials
  u32x4* Wzb1 = reinterpret_cast<u32x4*>(ws.take<unsigned char>(qsplit_wzb_bytes(s)));
  u32x4* Wzb2 = reinterpret_cast<u32x4*>(ws.take<unsigned char>(qsplit_wzb_bytes(s)));

  // saved tensors
  Carver sv(const_cast<float*>(saved));
  const float* xT_own = sv.take<float>((size_t)M * F);
  const float* xT = xt_in ? x : xT_own;    // (X_TRANSPOSED: the caller's [B*K][F] copy; the forward left saved's own area unused)
  const float* maps[kCinMaxL];
  const float *tailY = nullptr, *tailUz = nullptr, *tailWsum = nullptr;
  const float *qtR = nullptr, *qtT = nullptr, *qtWsumL = nullptr, *qtCvec = nullptr, *qtWsumP = nullptr, *qtWsnP = nullptr, *qtWzT = nullptr;
  if (qtail) {   // saved layout of the quadratic tail: xT | map 0 | R | T | wsum_L | cvec
    maps[0] = sv.take<float>((size_t)M * s.HS(0));
    const float* q = sv.take<float>(qtail_saved_floats(s));
    qtR = q;
    qtT = qtR + (size_t)M * s.HS(0);
    qtWsumL = qtT + (size_t)F * F * H[0];
    qtCvec = qtWsumL + (size_t)H[1] * F;
    qtWsumP = qtCvec + 128;
    qtWsnP = qtWsumP + (size_t)H[0] * F;
    qtWzT = qtWsnP + qtail_wsn_floats(s);
  } else if (tail) {   // saved layout of the fused tail: xT | maps 0..L-3 | Y | Uz | wsum_L
    for (int l = 0; l < tg.p; ++l) maps[l] = sv.take<float>((size_t)M * s.HS(l));
    tailY = sv.take<float>((size_t)M * tg.JP);
    tailUz = sv.take<float>(tg.uz_floats + tg.uf_floats);
    tailWsum = sv.take<float>((size_t)tg.Hq * F);   // (bmT layout: [f][n])
  } else {
    for (int l = 0; l + 1 < L; ++l) maps[l] = sv.take<float>((size_t)M * s.HS(l));
  }

  // ---- head backward: dP, ddense_w, ddense_b
  const float* dPsrc = g;  // output_dim != 1: g is already dL/dpooled
  if (output_dim == 1 && qmerge) {
    dPsrc = dP;   // (merged launches: the head's backward rides in cin_qtail_xe_kernel / cin_reduce_expand_q_kernel, below)
  } else if (output_dim == 1) {
    ProfScope ps("cin_head_bwd", st);
    hipLaunchKernelGGL(cin_head_bwd_kernel, dim3(nblk), dim3(256), 0, st, g, dense_w, pooled, dP, small, B, (int)LK, kHeadChunk);
    // (fused tail: the fixed-order sum of the head's partials rides in the tail's first launch, below)
    // (fused / quadratic tail: the fixed-order sum of the head's partials rides in the tail's first launch, below)
    if (!tail) hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv((int)LK + 1, 64)), dim3(256), 0, st, small, ddense_w, (long)(LK + 1), nblk, ddense_b, (long)LK);
    FIL_CHECK_LAUNCH();
    dPsrc = dP;
  }
  if (!tail) ready(L);

  int cur = 0;
  int ltop = L - 1;          // first layer handled by the general kernels
  bool dx_started = false;   // has dxT been initialised yet
  bool have_gx0 = false;     // did a general layer-1 kernel produce Gx^0
  bool wz_prepacked = false; // fused tail with L == 3: layer 0's dZ weights were packed by the tail's first launch
  bool qm_joined = false;    // merged quadratic tail: the two-pass dZ launch left nothing but the linear term for the final transpose
  if (qmerge) {
    // ---- quadratic tail with merged weight gradients (cin_qmerge.h): [dW1 | dT] = pairs(x)^T [G1 | dP_L x1] in ONE launch; the data
    // gradients stay two launches of the pair-symmetric dZ kernel (G1 with W1, then the unscaled x1 with T, scaled by dP_L in the
    // final transpose).  G1 = dP_p S + dP_1 + dP_L R and the shortcut's dX part come out of cin_last_bwd2_kernel.
    const int p = tg.p, lL = L - 1, Hpp = tg.Hpp, Hq = tg.Hq, HS0 = s.HS(0);
    FIL_CHECK_ARG(bias && W[0] && W[p] && W[lL] && bias[p] && dW[0] && dW[p] && dW[lL] && dbias[0] && dbias[p] && dbias[lL]);
    const float* xpT = maps[p - 1];
    const int xps = s.xps(p);
    const float* dPp = dPsrc + (size_t)p * K;
    const float* dPL = dPsrc + (size_t)lL * K;
    const float* dPprev = dPsrc + (size_t)(p - 1) * K;
    const double algo_tail = gemm_flops(M, Hpp, F, Hq) + gemm_flops(M, Hq, F, tg.HL);   // the top two layers of the reference graph
    const double algo1 = gemm_flops(M, F, F, H[0]);
    const int symD = F / 2 + 1, Cl = F * symD;
    const int JTs = cin_jt_sym(F);
    const int periods = cdiv(F, cin_dz_h_per_period(JTs));
    const int tiles0 = periods * cin_dz_tiles_per_period(JTs) + 1;
    // Gbuf[1] is free (the layer loop does not run): xe | gxR | dxR
    const int XE = F + 3;
    float* xe = Gbuf[1];                         // [M][F+3]: x | 1 | dP_L | dP_p
    float* gxR = xe + (size_t)M * XE;
    float* dxR = gxR + (size_t)M * F;
    {
      ProfScope ps("cin_tail_a", st, (double)M * (2 * F + 3) * sizeof(float));
      const size_t sh = (size_t)256 * (F + 3 + (output_dim == 1 ? L : 0)) * sizeof(float);
      allow_lds(cin_qtail_xe_kernel, sh);
      const int np = (int)std::min<long>(((long)tiles0 * 32 * HS0 + 255) / 256, 1024);   // + W1 in the dZ kernel's slot order
      const int nq2 = qsplit ? 2 * std::min(cdiv(tiles0 * 512, 256), 256) : 0;           // + (split-bf16 mode) W1s and Ts in slot order as planes
      // (output_dim == 1: + the dense head's backward -- dP and the block partials of ddense_w | ddense_b)
      hipLaunchKernelGGL(cin_qtail_xe_kernel, dim3(qt_ndc + np + nq2), dim3(kXeThreads), sh, st, xT, dPL, dPp, (int)LK, K, xe, qt_dcpart, (int)M, F, qt_ndc,
                         output_dim == 1 ? g : nullptr, dense_w, pooled, dP, qt_hpart, (int)LK, lL, p, W[0], Wz, H[0], JTs, HS0, tiles0, np, qtT, Hpp,
                         Wzb1, Wzb2);
    }
    FIL_CHECK_LAUNCH();
    {
      ProfScope ps("cin_last_bwd", st, 6.0 * (double)M * Hpp * F);
      cin_launch_last_bwd2(st, JT, xT, xpT, xps, qtWsumP, qtWsnP, dPp, (int)LK, dPprev, Gbuf[cur], HS0, dxT, (int)M, F, K, Hpp, qtR, HS0, dPL, small);
    }
    FIL_CHECK_LAUNCH();
    int dw_parts = 0;   // partials [C][256] the weight-gradient launch leaves
    {
      ProfScope ps("cin_bwd_dw_q", st, algo1 + algo_tail, gemm_flops(M, 1, Cl, H[0]) + gemm_flops(M, 1, Cl, Hpp));
      if (qsplit) {   // split-bf16 operands (cin_qsplit.h): one partial per row split
        const DwqbPlan bp = cin_dwqb_plan(M, Cl + F, cu_count());
        cin_launch_dwq_b(st, bp, Gbuf[cur], xpT, HS0, xe, XE, part, (int)M, F, symD);
        dw_parts = bp.splits;
      } else {
        const bool f4 = knobs().dwfold4 != 0;
        const DwqPlan dp = f4 ? cin_dwq_plan4(M, Cl + F, cu_count()) : cin_dwq_plan(M, Cl + F, cu_count());
        if (f4)
          hipLaunchKernelGGL((cin_dwq_kernel<kDwqDepth, 4>), dim3((dp.wgs + 7) / 8 * 8), dim3(kDwqThreads), 0, st, Gbuf[cur], xpT, HS0, xe, XE, part, (int)M, F,
                             symD, dp.rows_per_split, dp.splits, dp.ncol_full, dp.rem, dp.wgs_full, dp.wgs);
        else
          hipLaunchKernelGGL((cin_dwq_kernel<kDwqDepth, 2>), dim3((dp.wgs + 7) / 8 * 8), dim3(kDwqThreads), 0, st, Gbuf[cur], xpT, HS0, xe, XE, part, (int)M, F,
                             symD, dp.rows_per_split, dp.splits, dp.ncol_full, dp.rem, dp.wgs_full, dp.wgs);
        dw_parts = dp.pairs;
      }
    }
    FIL_CHECK_LAUNCH();
    {
      // fixed-order sums of the partials -> dW1 (both rows of a pair), dT, v^T; + dbias1 from the column sums cin_last_bwd2_kernel left
      ProfScope ps("cin_reduce_dw", st);
      const int nb1 = cdiv(H[0], 64);
      const int nh = output_dim == 1 ? cdiv((int)LK + 1, 64) : 0;   // + ddense_w | ddense_b from the head's block partials
      hipLaunchKernelGGL(cin_reduce_expand_q_kernel, dim3((Cl + F + 3) / 4 + nb1 + nh), dim3(256), 0, st, part, dw_parts, F, symD, H[0], Hpp, dW[0], qt_dT,
                         vlast, small, ncol, dbias[0], nb1, qt_hpart, qt_ndc, (int)LK, ddense_w, ddense_b);
    }
    FIL_CHECK_LAUNCH();
    ready(L);
    ready(0);
    {
      ProfScope ps("cin_tail_params", st);
      const size_t sh = cin_qtail_params_lds(F, Hq);
      allow_lds(cin_qtail_params_kernel, sh);
      float* dcfin = qt_dcpart + (size_t)qt_ndc * kQtConst;
      hipLaunchKernelGGL(cin_qtail_params_kernel, dim3(2 * Hpp + 1), dim3(256), sh, st, W[p], qtWsumL, qt_dT, vlast, dW[p], part, Hpp, F, Hq, qt_dcpart, qt_ndc,
                         dcfin);
      hipLaunchKernelGGL(cin_qtail_fill_kernel, dim3(cdiv(Hq * F, kQtFillCols)), dim3(kQtFillThreads), 0, st, part, Hpp, dcfin, bias[p], qtWsumL, dW[lL], dbias[p], dbias[lL], F, Hq,
                         tg.HL);
    }
    FIL_CHECK_LAUNCH();
    ready(lL);
    ready(p);
    const int NHMAX = HS0 / 2;
    if (NHMAX == 64 && knobs().dz2 != 0) {
      // both data-gradient passes in one launch (cin_dz2_kernel): G1 with W1, then dP_L x1 with T, into one dX image
      ProfScope ps("cin_bwd_dz_q", st, algo1 + algo_tail, gemm_flops(M, 1, Cl, H[0]) + gemm_flops(M, 1, Cl, Hpp));
      // (the kernel also finishes dx: + the shortcut's part in dxT, + dP_L c, transposed to [B,F,K] on the way out)
      bool split_done = false;
      if (qsplit) {   // split-bf16 operands (the planes of both layers' slot-ordered weights came out of the operand-row launch)
        split_done = cin_launch_dz2_b(st, JTs, Gbuf[cur], xpT, HS0, dPL, (int)LK, K, Wzb1, Wzb2, xT, dxT, /*accumulate=*/1, (int)M, F, H[0], Hpp, periods, dx, qtCvec);
      }
      if (!split_done &&
          !cin_launch_dz2(st, JTs, Gbuf[cur], xpT, HS0, dPL, (int)LK, K, Wz, qtWzT, xT, dxT, /*accumulate=*/1, (int)M, F, H[0], Hpp, periods, dx, qtCvec))
        return fail(FIL_ERR_UNSUPPORTED, "fil_cin_bwd: no two-pass data-gradient kernel for JT=%d (F=%d)", JTs, F);
      qm_joined = true;
    } else {
      const bool two_waves = NHMAX == 64 && cin_dzs_two_waves(JTs) && tune.mb_forced != 2 && knobs().dzs_mb != 2;
      const int MBs = two_waves ? 1 : tune.mb_rows(M);
      const int ks = MBs != 1 ? 1 : tune.ksplit(M);
      const dim3 zgrid(ks == 4 ? cdiv((int)M, 32) : cdiv((int)M, 128 * MBs));
      {
        ProfScope ps("cin_bwd_dz_tail", st, algo_tail, gemm_flops(M, 1, Cl, Hpp));   // (T in slot order: packed and saved by the forward)
        cin_launch_dz3_sym(st, MBs, JTs, NHMAX, zgrid, xpT, HS0, qtWzT, xT, gxR, dxR, 0, (int)M, F, Hpp, periods, ks);
        // (gxR + dxR, scaled by dP_L, and the linear term join dX in the final transpose)
      }
      FIL_CHECK_LAUNCH();
      {
        ProfScope ps("cin_bwd_dz_l1", st, algo1, gemm_flops(M, 1, Cl, H[0]));
        cin_launch_dz3_sym(st, MBs, JTs, NHMAX, zgrid, Gbuf[cur], HS0, Wz, xT, gx0T, dxT, 1, (int)M, F, H[0], periods, ks);
      }
      have_gx0 = true;
    }
    FIL_CHECK_LAUNCH();
    dx_started = true;
    ltop = -1;
  } else if (qtail) {
    // ---- quadratic tail (cin_qtail.h).  pool_p goes back through the pooled-weights shortcut of layer p; pool_L = <x1, R> through the
    // first layer's pair-symmetric dW / dZ kernels with x1 (unscaled) as their "gradient" operand: dT = (pairs of x, one factor scaled
    // by dP_L)^T x1, the two halves of d<x1,R>/dx come out per row and are scaled by dP_L afterwards; G^{p-1} += dP_L R is elementwise.
    const int p = tg.p, lL = L - 1, Hpp = tg.Hpp, Hq = tg.Hq, HS0 = s.HS(0);
    FIL_CHECK_ARG(bias && W[p] && W[lL] && bias[p] && dW[p] && dW[lL] && dbias[p] && dbias[lL]);
    const float* xpT = maps[p - 1];
    const int xps = s.xps(p);
    const float* dPp = dPsrc + (size_t)p * K;
    const float* dPL = dPsrc + (size_t)lL * K;
    const float* dPprev = dPsrc + (size_t)(p - 1) * K;
    const double algo = gemm_flops(M, Hpp, F, Hq) + gemm_flops(M, Hq, F, tg.HL);
    // Gbuf[1] is free until the layer loop is over (L == 3: the loop runs layer 0 only): xs | gxR | dxR
    float* xs = Gbuf[1];                         // [M][F+1]: dP_L x | dP_p
    float* gxR = xs + (size_t)M * (F + 1);
    float* dxR = gxR + (size_t)M * F;
    {
      ProfScope ps("cin_tail_a", st, (double)M * (F + 64) * sizeof(float));
      const size_t sh = (size_t)256 * (F + 3) * sizeof(float);
      allow_lds(cin_qtail_scale_kernel, sh);
      const int nh = output_dim == 1 ? cdiv((int)LK + 1, 64) : 0;   // + the head's partial sums (as in the fused tail's first launch)
      // ... and the first layer's dZ weights in slot order (the packed-W buffer is idle until that layer's dZ kernel)
      FIL_CHECK_ARG(W[0]);
      const int JTs0 = cin_jt_sym(F);
      const int tiles0 = cdiv(F, cin_dz_h_per_period(JTs0)) * cin_dz_tiles_per_period(JTs0) + 1;
      const int np = (int)std::min<long>(((long)tiles0 * 32 * HS0 + 255) / 256, 1024);
      hipLaunchKernelGGL(cin_qtail_scale_kernel, dim3(qt_ndc + nh + np), dim3(256), sh, st, xT, dPL, dPp, (int)LK, K, xs, qt_dcpart, (int)M, F, qt_ndc, small,
                         ddense_w, ddense_b, (int)LK, nblk, nh, W[0], Wz, H[0], JTs0, HS0, tiles0);
      wz_prepacked = true;
    }
    FIL_CHECK_LAUNCH();
    ready(L);
    {
      // pooled-weights shortcut of layer p: G^{p-1} = dP_p S + dP_{p-1} (+ dP_L R), dX = dP_p x1 wsum_p
      ProfScope ps("cin_last_bwd", st, 6.0 * (double)M * Hpp * F);
      // (wsum_p and its operand copy were saved by the forward; + dP_L R on the way out: the pool_L part of G^{p-1})
      // (... and the partial column sums of G^{p-1} for dbias_{p-1}: `small` is free again, the head's partials were reduced above)
      cin_launch_last_bwd2(st, JT, xT, xpT, xps, qtWsumP, qtWsnP, dPp, (int)LK, dPprev, Gbuf[cur], HS0, dxT, (int)M, F, K, Hpp, qtR, HS0, dPL, small);
    }
    FIL_CHECK_LAUNCH();
    const int symD = F / 2 + 1, Cl = F * symD;
    const int JTs = cin_jt_sym(F);
    {
      ProfScope ps("cin_bwd_dw_tail", st, algo, gemm_flops(M, 1, Cl, Hpp));
      // F extra channel rows behind the pairs: v^T[f][h] = sum_m dP_p[m] x[m,f] x1[m,h], the rank-one part of dW_p (column F of xs)
      const int parts = launch_dw3(st, dw_plan(M, Cl + F, Hpp), xpT, HS0, xT, xs, F + 1, part, M, F, F, Hpp, symD, /*xtra=*/F);
      const long nW = (long)Cl * Hpp, pstride = (long)(Cl + F) * Hpp;
      hipLaunchKernelGGL(cin_reduce_expand_sym_kernel, dim3((int)((nW + 63) / 64) + cdiv(F * Hpp, 64)), dim3(256), 0, st, part, qt_dT, F, symD, Hpp, parts,
                         pstride, vlast, (long)F * Hpp);
    }
    FIL_CHECK_LAUNCH();
    {
      ProfScope ps("cin_tail_params", st);
      const size_t sh = cin_qtail_params_lds(F, Hq);   // (at least the 4 x 64 floats the extra workgroup folds the dc partials through)
      allow_lds(cin_qtail_params_kernel, sh);
      float* dcfin = qt_dcpart + (size_t)qt_ndc * kQtConst;
      hipLaunchKernelGGL(cin_qtail_params_kernel, dim3(2 * Hpp + 1), dim3(256), sh, st, W[p], qtWsumL, qt_dT, vlast, dW[p], part, Hpp, F, Hq, qt_dcpart, qt_ndc,
                         dcfin);
      hipLaunchKernelGGL(cin_qtail_fill_kernel, dim3(cdiv(Hq * F, kQtFillCols)), dim3(kQtFillThreads), 0, st, part, Hpp, dcfin, bias[p], qtWsumL, dW[lL], dbias[p], dbias[lL], F, Hq,
                         tg.HL);
    }
    FIL_CHECK_LAUNCH();
    ready(lL);
    ready(p);
    {
      const int periods = cdiv(F, cin_dz_h_per_period(JTs));
      ProfScope ps("cin_bwd_dz_tail", st, algo, gemm_flops(M, 1, Cl, Hpp));   // (T in slot order: packed and saved by the forward)
      const int NHMAX = HS0 / 2;
      const bool two_waves = NHMAX == 64 && cin_dzs_two_waves(JTs) && tune.mb_forced != 2 && knobs().dzs_mb != 2;
      const int MBs = two_waves ? 1 : tune.mb_rows(M);
      const int ks = MBs != 1 ? 1 : tune.ksplit(M);
      cin_launch_dz3_sym(st, MBs, JTs, NHMAX, dim3(ks == 4 ? cdiv((int)M, 32) : cdiv((int)M, 128 * MBs)), xpT, HS0, qtWzT, xT, gxR, dxR, 0, (int)M, F, Hpp,
                         periods, ks);
      // (gxR + dxR, scaled by dP_L, and the linear term join dX in the final transpose)
    }
    FIL_CHECK_LAUNCH();
    dx_started = true;
    ltop = p - 1;
  } else if (tail) {
    // ---- fused tail: both top layers' parameter gradients from Q = Z_p^T A (F+2 columns), data gradients from A Ueff^T
    const int p = tg.p, lL = L - 1;
    FIL_CHECK_ARG(bias && W[p] && W[lL] && bias[p] && dW[p] && dW[lL] && dbias[p] && dbias[lL]);
    const float* xpT = maps[p - 1];
    const int xps = s.xps(p);
    const double algo = gemm_flops(M, tg.Hpp, F, tg.Hq) + gemm_flops(M, tg.Hq, F, tg.HL);   // the two layers of the reference graph
    float* Apk = Gbuf[1];
    // the first general layer below the tail is the pair-symmetric layer 0 (L == 3): its slot-ordered weights can be packed now
    // (nothing else uses the packed-W buffer any more), together with A and the head's partial sums -- one launch for the three
    wz_prepacked = p == 1 && tune.sym && F >= 2;
    {
      ProfScope ps("cin_tail_a", st, (double)M * (F + 64) * sizeof(float));
      const int na = (int)std::min<long>((M * 16 * tg.NCB + 255) / 256, 4096);
      const int nh = output_dim == 1 ? cdiv((int)LK + 1, 64) : 0;
      int np = 0, JTs = 0, tiles0 = 0;
      if (wz_prepacked) {
        FIL_CHECK_ARG(W[0]);
        JTs = cin_jt_sym(F);
        tiles0 = cdiv(F, cin_dz_h_per_period(JTs)) * cin_dz_tiles_per_period(JTs) + 1;
        np = (int)std::min<long>(((long)tiles0 * 32 * s.HS(0) + 255) / 256, 1024);
      }
      hipLaunchKernelGGL(cin_tail_pre_kernel, dim3(na + nh + np), dim3(256), 0, st, xT, dPsrc, (int)LK, K, p, lL, Apk, (int)M, F, tg.NCB, na, small,
                         ddense_w, ddense_b, (int)LK, nblk, nh, W[0], Wz, H[0], JTs, s.HS(0), tiles0);
    }
    FIL_CHECK_LAUNCH();
    ready(L);
    const TailDwPlan tp = tail_dw_plan(M, tg.C1);
    {
      ProfScope ps("cin_bwd_dw_tail", st, algo, 2.0 * (double)M * tg.C1 * (F + 2));
      TailDwArgs a{Apk, xT, xpT, xps, part, (int)M, F, tg.Hpp, tg.JP, tp.rows_per_split, tp.blocks_x, tp.blocks_x * tp.splits, knobs().tail_settle != 0};
      cin_launch_tail_dw(st, tg.NCB, a);
    }
    FIL_CHECK_LAUNCH();
    {
      ProfScope ps("cin_tail_params", st);
      const int nblk_p = cdiv(tg.Cp, kTailPc);
      const int ldb = (tg.Hq + 3) & ~3;
      const size_t sh = ((size_t)(F + 1) * ldb + (size_t)tg.JP * (kTailPc + 4) + (size_t)kTailPc * ldb) * sizeof(float);
      allow_lds(cin_tail_params_kernel, sh);
      // the dwsum_L partials go behind the Q partials in `part`; the reduced ones row of Q into the (idle) v buffer of the last-layer shortcut
      float* partB = part + (size_t)tp.splits * tg.C1 * tg.JP;
      float* qones = vlast;
      hipLaunchKernelGGL(cin_tail_params_kernel, dim3(2 * nblk_p), dim3(256), sh, st, part, tp.splits, W[p], tailWsum, dW[p], dbias[p], partB, qones,
                         tg.Cp, F, tg.Hq, tg.JP);
      hipLaunchKernelGGL(cin_tail_fill_kernel, dim3(cdiv(tg.Hq * F, 64)), dim3(256), 0, st, partB, nblk_p, qones, bias[p], dW[lL], dbias[lL], F,
                         tg.Hq, tg.HL);
    }
    FIL_CHECK_LAUNCH();
    ready(lL);
    ready(p);
    {
      ProfScope ps("cin_bwd_dz_tail", st, algo, 2.0 * (double)M * tg.Cp * (F + 1));
      TailDzArgs a{tailUz, xT, xpT, xps, tailY, tg.JP, dPsrc, (int)LK, K, p, lL, Gbuf[cur], s.HS(p - 1), dxT, (int)M, F, tg.Hpp, tg.periods, knobs().tail_dz_mode,
                   tune.ksplit(M)};
      cin_launch_tail_dz(st, JT, tg.NQ, a);
    }
    FIL_CHECK_LAUNCH();
    dx_started = true;
    ltop = p - 1;
  } else if (mode == 0) {
    // ---- last layer through the pooled-weights shortcut (see cin_last_* kernels)
    const int l = L - 1;
    FIL_CHECK_ARG(W[l] && dW[l] && dbias[l]);
    const int Hp = s.Hp(l), Hl = H[l], xps = s.xps(l);
    const float* xpT = l == 0 ? xT : maps[l - 1];
    const float* dPl = dPsrc + (size_t)l * K;
    const float* dPprev = l > 0 ? dPsrc + (size_t)(l - 1) * K : nullptr;
    const size_t shw = (size_t)Hp * ((F + 3) & ~3) * sizeof(float);
    ProfScope ps("cin_last_bwd", st, 6.0 * (double)M * Hp * F);
    // (l > 0: wsum also in the MFMA operand layout of cin_last_bwd2_kernel -- the dZ kernels' packed-W buffer is idle until
    // the first general layer packs into it, and nothing between here and that kernel touches it)
    hipLaunchKernelGGL(cin_wsum_wsn_kernel, dim3(cdiv(Hp * F, 8)), dim3(256), 0, st, W[l], wsum, Hp * F, Hl, l > 0 ? Wz : nullptr, Hp, F, 2 * JT,
                       chunks_of(Hp));
    hipLaunchKernelGGL(cin_slice_sum_kernel, dim3(nblk), dim3(256), 0, st, dPl, (int)LK, small, B, K, kHeadChunk);
    // dW_L[c,:] = v[c],  v[h,f] = sum_m x^{L-1}[m,h] * (x[m,f] dP[m]): the weight-gradient kernel with a single
    // all-ones field (F' = 1, so c = h) and G' = x * dP ([M][128], zero padded) as its right-hand side
    {
      float* yT = Gbuf[1];
      const int YS = (F + 3) & ~3;
      const long tot = M * YS;
      hipLaunchKernelGGL(cin_scale_rows3_kernel, dim3((int)std::min<long>((tot + 255) / 256, 4096)), dim3(256), 0, st, xT, dPl, (int)LK, K, yT, (int)M, F, YS);
      // l > 0: the roles are swapped -- the F fields of G' are the channel rows and x^{L-1} ([M][HS], 128-aligned rows) is the
      // streamed right-hand side, so the kernel computes v^T [F][Hp]: 2 channel blocks x a full 128-column chunk instead of
      // 4 blocks x a chunk that is 70 % padding (27 -> 14 us at c4); the fill kernel reads it transposed
      const bool swap = l > 0;
      int nb;
      if (swap) nb = launch_dw3(st, dw_plan(M, F, Hp), xpT, xps, nullptr, yT, YS, part, M, /*F=*/1, /*Hp=*/F, /*H=*/Hp);
      else nb = launch_dw3(st, dw_plan(M, Hp, F), yT, YS, nullptr, xpT, xps, part, M, /*F=*/1, Hp, /*H=*/F);
      hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv((int)cl, 64)), dim3(256), 0, st, part, vlast, (long)cl, nb);
      // (the same launch finishes dbias_L from the slice sums above)
      hipLaunchKernelGGL(cin_fill_rows_kernel, dim3((int)std::min<long>(((long)cl * Hl + 255) / 256, 2048)), dim3(256), 0, st, vlast, dW[l], (long)cl, Hl,
                         swap ? F : 0, Hp, small, nblk, dbias[l]);
    }
    ready(l);
    // G^{L-1} and dX
    if (l > 0) {
      cin_launch_last_bwd2(st, JT, xT, xpT, xps, wsum, Wz, dPl, (int)LK, dPprev, Gbuf[cur], s.HS(l - 1), dxT, (int)M, F, K, Hp);
    } else {
      const size_t shb = shw + (size_t)kLastRows * (kLastFMax + 1) * sizeof(float);
      allow_lds(cin_last_bwd_kernel, shb);
      hipLaunchKernelGGL(cin_last_bwd_kernel, dim3(cdiv((int)M, kLastRows)), dim3(256), shb, st, xT, xpT, xps, wsum, dPl, (int)LK, dPprev,
                         nullptr, 0, dxT, /*layer1=*/1, (int)M, F, K, Hp);
    }
    FIL_CHECK_LAUNCH();
    dx_started = true;
    ltop = L - 2;
  } else {
    // ---- top layer gradient: broadcast of its pooled gradient
    const long total = M * s.HS(L - 1);
    ProfScope ps("cin_bcast_g", st, (double)total * sizeof(float));
    hipLaunchKernelGGL(cin_bcast3_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, dPsrc + (size_t)(L - 1) * K, (int)LK,
                       K, Gbuf[cur], s.HS(L - 1), (int)M, H[L - 1]);
    FIL_CHECK_LAUNCH();
  }
  for (int l = ltop; l >= 0; --l) {
    FIL_CHECK_ARG(W[l] && dW[l] && dbias[l]);
    const int Hp = s.Hp(l), Hl = H[l], HSl = s.HS(l), xps = s.xps(l);
    const float* xpT = l == 0 ? xT : maps[l - 1];
    const float* G = Gbuf[cur];
    // dbias
    {
      ProfScope ps("cin_dbias", st, (double)M * Hl * sizeof(float));
      // (quadratic tail: the kernel that wrote G left its per-128-row column sums in `small` already)
      if (!(qtail && l == ltop)) hipLaunchKernelGGL(cin_colsum3_kernel, dim3(ncol), dim3(256), 0, st, G, HSl, small, (int)M, Hl, kColRows);
      hipLaunchKernelGGL(cin_reduce_kernel, dim3(cdiv(Hl, 64)), dim3(256), 0, st, small, dbias[l], (long)Hl, ncol);
    }
    FIL_CHECK_LAUNCH();
    // dW
    int parts;
    const int symD = (l == 0 && tune.sym) ? F / 2 + 1 : 0;   // unordered field pairs: half the channels
    const int Cl = symD > 0 ? F * symD : Hp * F;
    {
      ProfScope ps(kDwNames[l], st, gemm_flops(M, Hp, F, Hl), gemm_flops(M, 1, Cl, Hl));
      parts = launch_dw3(st, dw_plan(M, Cl, Hl), G, HSl, xT, xpT, xps, part, M, F, Hp, Hl, symD);
    }
    FIL_CHECK_LAUNCH();
    const long nW = (long)Cl * Hl;
    {
      ProfScope ps("cin_reduce_dw", st, (double)(parts + 1) * nW * sizeof(float));
      // (pair-indexed first layer: the fixed-order sum is written straight to both dW rows of each pair)
      if (symD > 0) hipLaunchKernelGGL(cin_reduce_expand_sym_kernel, dim3((int)((nW + 63) / 64)), dim3(256), 0, st, part, dW[l], F, symD, Hl, parts);
      else hipLaunchKernelGGL(cin_reduce_kernel, dim3((int)((nW + 63) / 64)), dim3(256), 0, st, part, dW[l], nW, parts);
    }
    FIL_CHECK_LAUNCH();
    ready(l);   // dW[l], dbias[l] final: the dZ kernel of this layer and everything below can overlap their all-reduce
    // dZ -> G^{l-1}, dX
    {
      const int NHMAX = HSl / 2;                  // 64 (H <= 128) or 128
      const int MB = NHMAX == 128 ? 1 : tune.mb_rows(M);      // pair-symmetric first layer: 64 rows per wave measured best
      // general dZ kernel: 32 rows per wave, two waves per SIMD -- the second wave covers the issue time of the first
      // one's register contraction (c4: 0.715 -> 0.687 ms); FIL_CIN_DZ_MB overrides
      const int MBg = NHMAX == 128 ? 1 : ((knobs().dz_mb == 2 && JT <= 28) ? 2 : 1);   // (JT = 32 at 64 rows: scratch + line buffers > 160 KB of LDS)
      const float* dPprev = l > 0 ? dPsrc + (size_t)(l - 1) * K : nullptr;
      if (l == 0 && tune.sym && F >= 2) {
        // first layer over unordered field pairs (half the tiles); F = 1 would make both lane halves hit one word
        const int JTs = cin_jt_sym(F);
        const int periods = cdiv(F, cin_dz_h_per_period(JTs));
        const int tiles = periods * cin_dz_tiles_per_period(JTs) + 1;
        const long npack = (long)tiles * 32 * HSl;
        // exact kernel, H <= 128: 32 rows per wave at TWO waves per SIMD where the instantiation fits 256 registers (the second wave
        // covers the first one's prologue, epilogue and contraction issue: c4 0.154 -> 0.136 ms); FIL_CIN_MB2 / FIL_CIN_MB=2 /
        // FIL_CIN_DZS_MB=2 keep 64 rows per wave
        const bool two_waves = NHMAX == 64 && cin_dzs_two_waves(JTs) && tune.mb_forced != 2 && knobs().dzs_mb != 2;
        const int MBs = two_waves ? 1 : MB;
        if (!wz_prepacked) {
          hipLaunchKernelGGL(cin_pack_wz_sym_kernel, dim3((int)std::min<long>((npack + 255) / 256, 2048)), dim3(256), 0, st, W[l], Wz, F, Hl, JTs, HSl, tiles);
        }
        ProfScope ps(kDzNames[l], st, gemm_flops(M, Hp, F, Hl), gemm_flops(M, 1, F * (F / 2 + 1), Hl));
        const int ks = (NHMAX != 64 || MBs != 1) ? 1 : tune.ksplit(M);
        cin_launch_dz3_sym(st, MBs, JTs, NHMAX, dim3(ks == 4 ? cdiv((int)M, 32) : cdiv((int)M, 128 * MBs)), G, HSl, Wz, xT, gx0T, dxT,
                           dx_started ? 1 : 0, (int)M, F, Hl, periods, ks);
      } else {
        const int periods = dz_periods(s, l);
        const int tiles = periods * cin_dz_tiles_per_period(JT) + 1;
        const long npack = (long)tiles * 32 * HSl;
        hipLaunchKernelGGL(cin_pack_wz_kernel, dim3((int)std::min<long>((npack + 255) / 256, 2048)), dim3(256), 0, st, W[l], Wz, Hp, F, Hl, JT, HSl, tiles);
        ProfScope ps(kDzNames[l], st, gemm_flops(M, Hp, F, Hl));
        cin_launch_dz3(st, MBg, JT, NHMAX, dim3(cdiv((int)M, 128 * MBg)), G, HSl, Wz, xT, xpT, xps, dPprev, (int)LK, K,
                       l > 0 ? Gbuf[cur ^ 1] : nullptr, l > 0 ? s.HS(l - 1) : 0, l == 0 ? gx0T : nullptr, dxT, dx_started ? 1 : 0,
                       (int)M, F, Hp, Hl, periods);
      }
      dx_started = true;
      if (l == 0) have_gx0 = true;
    }
    FIL_CHECK_LAUNCH();
    cur ^= 1;
  }
  if (!qm_joined) {   // (merged quadratic tail: cin_dz2_kernel wrote dx itself)
    ProfScope ps("cin_transpose_out", st, 2.0 * M * F * sizeof(float));
    if (qmerge)
      hipLaunchKernelGGL(cin_transpose_out_kernel, dim3(B), dim3(256), (size_t)K * (F + 1) * sizeof(float), st, dxT, have_gx0 ? gx0T : nullptr, dx, F, K,
                         Gbuf[1] + (size_t)M * (F + 3), Gbuf[1] + (size_t)M * (F + 3) + (size_t)M * F, qtCvec, dPsrc + (size_t)(L - 1) * K, (int)LK);
    else if (qtail)
      hipLaunchKernelGGL(cin_transpose_out_kernel, dim3(B), dim3(256), (size_t)K * (F + 1) * sizeof(float), st, dxT, have_gx0 ? gx0T : nullptr, dx, F, K,
                         Gbuf[1] + (size_t)M * (F + 1), Gbuf[1] + (size_t)M * (F + 1) + (size_t)M * F, qtCvec, dPsrc + (size_t)(L - 1) * K, (int)LK);
    else
      hipLaunchKernelGGL(cin_transpose_out_kernel, dim3(B), dim3(256), (size_t)K * (F + 1) * sizeof(float), st, dxT, have_gx0 ? gx0T : nullptr, dx, F, K);
  }
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
