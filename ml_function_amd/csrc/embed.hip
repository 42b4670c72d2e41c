// N1  SparseEmbed field-index work for gfx950: packed [B,F,K] gather and its gradient.
//
// Replaces the F separate Embedding lookups + Concatenate of the reference
// (interactive_layer.py:225-242, models.py:131): one launch emits the packed layout the interaction
// kernels consume.  Rows are copied bit-exactly.  A (b,f) row of K floats is moved by K/4 lanes with
// 16-byte accesses; consecutive lanes walk consecutive rows of the output, so stores are fully coalesced
// and each gathered table row is read as one contiguous 4*K-byte segment.
//
// Ids are range-checked against the per-field vocabulary when `sizes` is given: an id outside [0, V_f) yields a zero row
// (what Keras' Embedding does on a GPU; on the CPU it raises -- the Python layer can ask for that, see oob_count) instead
// of silently reading the next field's table, and its gradient is dropped instead of corrupting that table.
//
// Gradient, default = deterministic: the caller sorts the global row ids (stable), the kernel sums each row's
// contributions in sorted order with a fixed lane tree (fil_embed_segment_sum) and either scatters the unique rows into
// the dense table or hands them out as (rows, values) -- no atomics, no dense zero table needed.  The fp32-atomic
// scatter-add (order of additions not fixed) stays available as fil_embed_scatter_add.
#include "common.h"
#include <hip/hip_bf16.h>
#include <cstdlib>

namespace fil {

// OT = __hip_bfloat16: the rows leave rounded to bf16 (the block a bf16 model would otherwise cast in a launch of its own)
template <int VEC, typename OT = float>
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ offsets,
                                                           const int64_t* __restrict__ sizes, const int64_t* __restrict__ idx,
                                                           OT* __restrict__ out, int* __restrict__ oob_count, long rows, int F,
                                                           int K) {
  const int KV = K / VEC;
  const long total = rows * KV;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long row = t / KV;
    const int kv = (int)(t - row * KV);
    const int f = (int)(row % F);
    const int64_t id = idx[row];
    const bool ok = sizes == nullptr || (id >= 0 && id < sizes[f]);
    if (!ok && kv == 0 && oob_count != nullptr) atomicAdd(oob_count, 1);
    const long src = (offsets[f] + (ok ? id : 0)) * K + (long)kv * VEC;
    if constexpr (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(table + src);
      if constexpr (sizeof(OT) == 4) {
        *reinterpret_cast<float4*>(out + row * K + kv * 4) = v;
      } else {
        OT* o = out + row * K + kv * 4;
        o[0] = (OT)v.x;
        o[1] = (OT)v.y;
        o[2] = (OT)v.z;
        o[3] = (OT)v.w;
      }
    } else {
      out[row * K + kv] = (OT)(ok ? table[src] : 0.f);
    }
  }
}

// The same gather, one workgroup per sample through LDS, writing BOTH layouts: the packed [B,F,K] block and the row-major
// transposed xT [B*K][F] (xT[(b*K + k)*F + f] = out[b,f,k]) that the CIN kernels consume -- the consumer's input transpose
// (cin_transpose_in: one more read and write of the block) disappears.  Both outputs leave as whole contiguous rows.
__global__ __launch_bounds__(256) void embed_gather_xt_kernel(const float* __restrict__ table, const int64_t* __restrict__ offsets,
                                                              const int64_t* __restrict__ sizes, const int64_t* __restrict__ idx,
                                                              float* __restrict__ out, float* __restrict__ out_t,
                                                              int* __restrict__ oob_count, int F, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [F][K+1]
  const long b = blockIdx.x;
  for (int i = threadIdx.x; i < F * K; i += 256) {
    const int f = i / K, k = i - f * K;
    const int64_t id = idx[b * F + f];
    const bool ok = sizes == nullptr || (id >= 0 && id < sizes[f]);
    if (!ok && k == 0 && oob_count != nullptr) atomicAdd(oob_count, 1);
    const float v = ok ? table[(offsets[f] + id) * K + k] : 0.f;
    smem[f * (K + 1) + k] = v;
    out[b * F * K + i] = v;
  }
  __syncthreads();
  float* dst = out_t + b * K * F;
  for (int i = threadIdx.x; i < F * K; i += 256) dst[i] = smem[(i % F) * (K + 1) + (i / F)];
}

// dtable[offsets[f] + idx[b,f], k] += g[b,f,k]  -- fp32 global atomics (one dword per lane, contiguous per row).
// The order of additions into a row that is hit several times in a batch is not fixed.
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ sizes,
                                                            const int64_t* __restrict__ idx, const float* __restrict__ g,
                                                            float* __restrict__ dtable, long rows, int F, int K) {
  const long total = rows * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long row = t / K;
    const int k = (int)(t - row * K);
    const int f = (int)(row % F);
    const int64_t id = idx[row];
    if (sizes != nullptr && (id < 0 || id >= sizes[f])) continue;
    atomicAdd(dtable + (offsets[f] + id) * K + k, g[t]);
  }
}

// global row id of every (b,f): offsets[f] + idx[b,f], or -1 for an out-of-range / frozen-field id (sorts first, skipped)
__global__ __launch_bounds__(256) void embed_row_ids_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ sizes,
                                                            const unsigned char* __restrict__ frozen, const int64_t* __restrict__ idx,
                                                            int64_t* __restrict__ row_ids, long rows, int F) {
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const int f = (int)(r % F);
    const int64_t id = idx[r];
    const bool ok = (sizes == nullptr || (id >= 0 && id < sizes[f])) && (frozen == nullptr || !frozen[f]);
    row_ids[r] = ok ? offsets[f] + id : -1;
  }
}

// The sort of the deterministic gradient without a library sort: the global row ids of DIFFERENT fields never collide (offsets[f]
// separate them), so equal ids only have to be adjacent WITHIN a field.  One workgroup per field sorts the field's B pairs
// (local id + 1 or 0 for a skipped entry, position b) as 64-bit composites -- unique, so the order among equal ids is by position,
// i.e. stable -- with a bitonic network in LDS, and writes them as the field's segment [f B, (f+1) B) of sorted_ids / perm
// (perm = b F + f, the flat position torch.sort would report).  fil_embed_run_sum only needs equal ids adjacent, in a fixed order.
// (torch.sort on 160 k keys is eight merge launches + casts: ~75 us of a 1.4 ms xDeepFM step; this is one 43 us launch -- 78 compare-exchange
// steps of 64 KB of LDS traffic and a 16-wave barrier each with 64-bit composites; 32-bit ones where the vocabulary allows)
// KeyT = unsigned: the composite is (id + 1) << log2(N) | position in 32 bits -- half the LDS traffic of a step -- when every id + 1
// fits the bits the positions leave (the host checks max_vocab); unsigned long long: (id + 1) << 32 | position.
template <typename KeyT>
__global__ __launch_bounds__(1024) void embed_sort_fields_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ sizes,
                                                                 const unsigned char* __restrict__ frozen, const int64_t* __restrict__ idx,
                                                                 int64_t* __restrict__ sorted_ids, int64_t* __restrict__ perm, int B, int F, int N,
                                                                 int pos_bits) {
  extern __shared__ unsigned long long skeys_raw[];
  KeyT* skeys = reinterpret_cast<KeyT*>(skeys_raw);
  const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const bool live = frozen == nullptr || !frozen[f];
  const int64_t vf = sizes != nullptr ? sizes[f] : (int64_t)0x7fffffff;
  for (int i = tid; i < N; i += nt) {
    KeyT c = ~(KeyT)0;            // padding sorts last
    if (i < B) {
      const int64_t id = idx[(long)i * F + f];
      const bool ok = live && id >= 0 && id < vf;
      c = ((KeyT)(ok ? (unsigned)id + 1u : 0u) << pos_bits) | (KeyT)(unsigned)i;
    }
    skeys[i] = c;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += nt) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const bool up = (i & k) == 0;
        const KeyT a = skeys[i], b = skeys[l];
        if ((a > b) == up) {
          skeys[i] = b;
          skeys[l] = a;
        }
      }
      __syncthreads();
    }
  }
  const int64_t off = offsets[f];
  const KeyT pos_mask = ((KeyT)1 << pos_bits) - 1;
  for (int i = tid; i < B; i += nt) {
    const KeyT c = skeys[i];
    const unsigned key = (unsigned)(c >> pos_bits);
    sorted_ids[(long)f * B + i] = key != 0u ? off + (int64_t)(key - 1u) : -1;
    perm[(long)f * B + i] = (int64_t)(unsigned)(c & pos_mask) * F + f;
  }
}

// The same sort with the keys in REGISTERS: thread t of 1024 owns the E = N / 1024 consecutive elements t E .. t E + E - 1.  A
// compare-exchange step at distance j < E is register-to-register, at E <= j < 64 E a lane exchange inside the wave (the partner of
// element i is element i ^ j: the same slot of thread t ^ (j / E)), and only the steps at distance >= 64 E cross waves through LDS --
// 10 of the 78 steps of N = 4096 (E = 4) need a barrier, against all 78 of the kernel above.  The composites are unique, so "the
// lower index keeps the smaller one" is the same network as the swap form.
template <typename KeyT>
__device__ __forceinline__ KeyT lane_xor(KeyT v, int m) {
  if constexpr (sizeof(KeyT) == 4) {
    return (KeyT)__shfl_xor((int)v, m, 64);
  } else {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
    return ((KeyT)hi << 32) | lo;
  }
}
template <typename KeyT, int E>
__global__ __launch_bounds__(1024) void embed_sort_fields_reg_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ sizes,
                                                                     const unsigned char* __restrict__ frozen, const int64_t* __restrict__ idx,
                                                                     int64_t* __restrict__ sorted_ids, int64_t* __restrict__ perm, int B, int F,
                                                                     int pos_bits) {
  constexpr int N = 1024 * E;
  __shared__ KeyT xch[N];
  const int f = blockIdx.x, t = threadIdx.x;
  const bool live = frozen == nullptr || !frozen[f];
  const int64_t vf = sizes != nullptr ? sizes[f] : (int64_t)0x7fffffff;
  KeyT v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    KeyT c = ~(KeyT)0;            // padding sorts last
    if (i < B) {
      const int64_t id = idx[(long)i * F + f];
      const bool ok = live && id >= 0 && id < vf;
      c = ((KeyT)(ok ? (unsigned)id + 1u : 0u) << pos_bits) | (KeyT)(unsigned)i;
    }
    v[e] = c;
  }
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if ((e & j) == 0) {
            const bool up = k < E ? (e & k) == 0 : ((t * E) & k) == 0;
            const KeyT a = v[e], b = v[e | j];
            const bool sw = (a > b) == up;
            v[e] = sw ? b : a;
            v[e | j] = sw ? a : b;
          }
        }
      } else {
        const int m = j / E;                       // partner thread t ^ m
        const bool up = ((t * E) & k) == 0;        // (k > j >= E: the same for the thread's E elements)
        const bool lower = (t & m) == 0;
        KeyT o[E];
        if (m < 64) {
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = lane_xor<KeyT>(v[e], m);
        } else {
          __syncthreads();                         // (the previous exchange's reads are done)
#pragma unroll
          for (int e = 0; e < E; ++e) xch[t * E + e] = v[e];
          __syncthreads();
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = xch[(t ^ m) * E + e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const KeyT lo = v[e] < o[e] ? v[e] : o[e], hi = v[e] < o[e] ? o[e] : v[e];
          v[e] = (lower == up) ? lo : hi;
        }
      }
    }
  }
  const int64_t off = offsets[f];
  const KeyT pos_mask = ((KeyT)1 << pos_bits) - 1;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = t * E + e;
    if (i < B) {
      const KeyT c = v[e];
      const unsigned key = (unsigned)(c >> pos_bits);
      sorted_ids[(long)f * B + i] = key != 0u ? off + (int64_t)(key - 1u) : -1;
      perm[(long)f * B + i] = (int64_t)(unsigned)(c & pos_mask) * F + f;
    }
  }
}

// One wave per unique row u: values[u,:] = sum_{j in [starts[u], starts[u+1])} g[perm[j], :], contributions taken in sorted
// order, C = 64 / KQ of them in parallel (lane = c * KQ + kq, kq = 4 consecutive k), folded with a fixed xor tree.
// rows_out[u] < 0 (the out-of-range bucket) is skipped.  dtable != NULL: the sum is also stored to dtable[rows_out[u], :]
// (each row written by exactly one wave: no atomics).
__global__ __launch_bounds__(256) void embed_segment_sum_kernel(const float* __restrict__ g, const int64_t* __restrict__ perm,
                                                                const int64_t* __restrict__ starts, const int64_t* __restrict__ rows_out,
                                                                float* __restrict__ values, float* __restrict__ dtable, long U, int K) {
  __shared__ float red[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KQ = (K + 3) / 4;                 // 4-wide pieces of a row (the last may be partial)
  const int C = 64 / KQ;                      // contributions in flight per wave (host guarantees KQ <= 64)
  const int c = lane / KQ, kq = lane - c * KQ;
  const bool live = c < C;
  for (long u = (long)blockIdx.x * 4 + wave; u < U; u += (long)gridDim.x * 4) {
    const int64_t row = rows_out[u];
    const long lo = starts[u], hi = starts[u + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (live && row >= 0) {
      for (long j = lo + c; j < hi; j += C) {
        const float* src = g + perm[j] * K + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kq * 4 + i < K) acc[i] += src[i];
      }
    }
    // fold the C partial sums of every kq in a fixed order: through LDS, lane (0, kq) adds c = 1 .. C-1 in turn
    float* my = red[wave];
#pragma unroll
    for (int i = 0; i < 4; ++i) my[lane * 4 + i] = acc[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c == 0 && row >= 0) {
      for (int cc = 1; cc < C; ++cc)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += my[(cc * KQ + kq) * 4 + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kq * 4 + i;
        if (k < K) {
          if (values != nullptr) values[u * K + k] = acc[i];
          if (dtable != nullptr) dtable[row * K + k] = acc[i];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Capture-safe form of the same sum (no data-dependent sizes anywhere).  A wave takes C = 64 / KQ consecutive SORTED POSITIONS
// per iteration, one per lane group (KQ lanes = one row of K floats).  A group whose position starts a run of equal row ids
//   * of at most kRunShort elements sums it by itself, in sorted order (the ids, the permutation entries and the gradient rows
//     of the whole run are three rounds of independent loads: Criteo-like batches are mostly runs of 1-3);
//   * of more elements hands it to the whole wave: lanes (c, kq) take elements c, c + C, ... and the C partial sums are folded
//     in lane order (hot ids with thousands of hits).
// Which form a run takes depends on its length only, so repeats are bit-identical.  Rows are stored into the zeroed dense
// table; id -1 (out-of-range / frozen) is skipped.  (Round 2 first had one wave per position: 60 us for 160 k positions.)
constexpr int kRunShort = 8;

template <typename GT>   // GT = __hip_bfloat16: the incoming gradient rows are bf16 (converted on load, summed in fp32)
__global__ __launch_bounds__(256) void embed_run_sum_kernel(const GT* __restrict__ g, const int64_t* __restrict__ perm,
                                                            const int64_t* __restrict__ sorted_ids, float* __restrict__ dtable, long R,
                                                            int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KQ = (K + 3) / 4, C = 64 / KQ;
  const int c = lane / KQ, kq = lane - c * KQ;
  for (long j0 = ((long)blockIdx.x * 4 + wave) * C; j0 < R; j0 += (long)gridDim.x * 4 * C) {
    const long j = j0 + c;
    const bool have = c < C && j < R;
    const long jc = have ? j : R - 1;
    const int64_t row = sorted_ids[jc];
    const int64_t prev = jc > 0 ? sorted_ids[jc - 1] : -2;
    const bool start = have && row >= 0 && prev != row;
    // length of the run, counted up to kRunShort + 1 (all look-ahead ids are independent loads)
    int64_t nxt[kRunShort];
#pragma unroll
    for (int t = 0; t < kRunShort; ++t) nxt[t] = sorted_ids[jc + 1 + t < R ? jc + 1 + t : R - 1];
    int n = 1;
    bool same = true;
#pragma unroll
    for (int t = 0; t < kRunShort; ++t) {
      same = same && jc + 1 + t < R && nxt[t] == row;
      n += same ? 1 : 0;
    }
    const bool is_long = start && n > kRunShort;
    if (start && !is_long) {
      long pr[kRunShort];
#pragma unroll
      for (int t = 0; t < kRunShort; ++t) pr[t] = perm[t < n ? jc + t : jc];
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < kRunShort; ++t) {
        const GT* src = g + pr[t] * K + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t < n && kq * 4 + i < K) acc[i] += (float)src[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (kq * 4 + i < K) dtable[row * K + kq * 4 + i] = acc[i];
    }
    // long runs: one after the other, by the whole wave
    unsigned long long pending = __ballot(is_long && kq == 0);
    while (pending != 0) {
      const int ll = __builtin_ctzll(pending);          // first lane of the group that found the run
      pending &= pending - 1;
      const long js = j0 + ll / KQ;
      // the run's id from the lane that found it (a reload was one more L2 round trip per run)
      const int64_t rl = ((int64_t)__shfl((int)(row >> 32), ll, 64) << 32) | (unsigned)__shfl((int)(unsigned)row, ll, 64);
      // The end of the run, 64 ids per look (one ballot), and -- requested together with the first look, before its answer -- the
      // permutation entries of the first 4 C elements: a run of up to 4 C elements (most of the "long" ones: 9 ... 64) is then two
      // dependent round trips (ids | perm, rows) instead of four.  The sums run over a KNOWN range (a compare per element would
      // hang every load on an id: a field of 10 values in a batch of 4096 is ten runs of ~400).
      const long jj0 = js + c;
      long pr0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long q = jj0 + (long)u * C;
        pr0[u] = perm[(c < C && q < R) ? q : js];
      }
      long je = js + 1;
      for (;;) {
        const long pp = je + lane;
        const unsigned long long same = __ballot(pp < R && sorted_ids[pp < R ? pp : R - 1] == rl);
        if (same == ~0ull) {
          je += 64;
          continue;
        }
        je += __builtin_ctzll(~same);
        break;
      }
      // lane group c takes elements js + c, + C, ...: four at a time into four accumulators (independent loads), folded as
      // (a0 + a1) + (a2 + a3) -- the order depends on the run's length only, so repeats stay bit-identical
      float acc4[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[u][i] = 0.f;
      if (c < C) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const GT* src = g + pr0[u] * K + kq * 4;
          const bool on = jj0 + (long)u * C < je;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (on && kq * 4 + i < K) acc4[u][i] += (float)src[i];
        }
        // (the next batch's permutation entries requested in front of this batch's rows: 62 -> 90 registers, 20.9 -> 21.9 us)
        for (long jj = jj0 + 4L * C; jj < je; jj += 4L * C) {
          long pr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) pr[u] = perm[jj + (long)u * C < je ? jj + (long)u * C : jj];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const GT* src = g + pr[u] * K + kq * 4;
            const bool on = jj + (long)u * C < je;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (on && kq * 4 + i < K) acc4[u][i] += (float)src[i];
          }
        }
      }
      float acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = (acc4[0][i] + acc4[1][i]) + (acc4[2][i] + acc4[3][i]);
      // the C partial sums meet in a fixed tree of lane exchanges (group c takes group c + s, s = 1, 2, 4, ...): through LDS, lane
      // group 0 adding the others one by one, the fold was C - 1 dependent LDS reads per run
      for (int sft = 1; sft < C; sft <<= 1) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __shfl(acc[i], (lane + sft * KQ) & 63, 64);
        if (c < C && (c & (2 * sft - 1)) == 0 && c + sft < C) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += o[i];
        }
      }
      if (c == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kq * 4 + i < K) dtable[rl * K + kq * 4 + i] = acc[i];
      }
    }
  }
}

}  // namespace fil

using namespace fil;

extern "C" int fil_embed_run_sum_dt(const void* g, const int64_t* perm, const int64_t* sorted_ids, float* dtable, long R, int K, int g_dtype,
                                    void* stream) {
  FIL_CHECK_ARG(R >= 0 && K >= 1);
  if (g_dtype != FIL_F32 && g_dtype != FIL_BF16) return fail(FIL_ERR_ARG, "fil_embed_run_sum_dt: g_dtype %d (f32 or bf16)", g_dtype);
  if (K > 256) return fail(FIL_ERR_UNSUPPORTED, "fil_embed_run_sum: K=%d > 256", K);
  if (R == 0) return FIL_OK;
  FIL_CHECK_ARG(g && perm && sorted_ids && dtable);
  const int C = 64 / ((K + 3) / 4);   // positions per wave and iteration
  const dim3 grid((int)std::min<long>((R + 4 * C - 1) / (4 * C), 256 * 32));
  if (g_dtype == FIL_F32)
    hipLaunchKernelGGL(embed_run_sum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(g), perm, sorted_ids, dtable, R, K);
  else
    hipLaunchKernelGGL(embed_run_sum_kernel<__hip_bfloat16>, grid, dim3(256), 0, (hipStream_t)stream, static_cast<const __hip_bfloat16*>(g), perm,
                       sorted_ids, dtable, R, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_run_sum(const float* g, const int64_t* perm, const int64_t* sorted_ids, float* dtable, long R, int K, void* stream) {
  return fil_embed_run_sum_dt(g, perm, sorted_ids, dtable, R, K, FIL_F32, stream);
}

extern "C" int fil_embed_gather_dt(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, void* out,
                                   int* oob_count, int B, int F, int K, int out_dtype, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (out_dtype != FIL_F32 && out_dtype != FIL_BF16) return fail(FIL_ERR_ARG, "fil_embed_gather_dt: out_dtype %d (f32 or bf16)", out_dtype);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(table && offsets && idx && out);
  const long rows = (long)B * F;
  const bool vec = (K % 4 == 0);
  const long total = rows * (vec ? K / 4 : K);
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 8);
  hipStream_t st = (hipStream_t)stream;
#define FIL_GATHER(VECV, OTV) \
  hipLaunchKernelGGL((embed_gather_kernel<VECV, OTV>), dim3(grid), dim3(256), 0, st, table, offsets, sizes, idx, static_cast<OTV*>(out), oob_count, rows, F, K)
  if (out_dtype == FIL_F32) {
    if (vec) FIL_GATHER(4, float); else FIL_GATHER(1, float);
  } else {
    if (vec) FIL_GATHER(4, __hip_bfloat16); else FIL_GATHER(1, __hip_bfloat16);
  }
#undef FIL_GATHER
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_gather(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, float* out,
                                int* oob_count, int B, int F, int K, void* stream) {
  return fil_embed_gather_dt(table, offsets, sizes, idx, out, oob_count, B, F, K, FIL_F32, stream);
}

extern "C" int fil_embed_gather_xt(const float* table, const int64_t* offsets, const int64_t* sizes, const int64_t* idx, float* out,
                                   float* out_t, int* oob_count, int B, int F, int K, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(table && offsets && idx && out && out_t);
  const size_t sh = (size_t)F * (K + 1) * sizeof(float);
  if (sh > 64 * 1024) return fail(FIL_ERR_UNSUPPORTED, "fil_embed_gather_xt: F*(K+1) = %d floats of LDS per sample (> 64 KiB)", F * (K + 1));
  if (sh > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(embed_gather_xt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FIL_ERR_HIP, "fil_embed_gather_xt: cannot reserve %zu bytes of LDS", sh);
  }
  hipLaunchKernelGGL(embed_gather_xt_kernel, dim3(B), dim3(256), sh, (hipStream_t)stream, table, offsets, sizes, idx, out, out_t, oob_count, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_scatter_add(const int64_t* offsets, const int64_t* sizes, const int64_t* idx, const float* g, float* dtable,
                                     int B, int F, int K, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && K >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(offsets && idx && g && dtable);
  const long rows = (long)B * F;
  const long total = rows * K;
  const int grid = (int)std::min<long>((total + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, offsets, sizes, idx, g, dtable, rows, F, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_row_ids(const int64_t* offsets, const int64_t* sizes, const unsigned char* frozen, const int64_t* idx,
                                 int64_t* row_ids, int B, int F, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(offsets && idx && row_ids);
  const long rows = (long)B * F;
  hipLaunchKernelGGL(embed_row_ids_kernel, dim3((int)std::min<long>((rows + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, offsets, sizes,
                     frozen, idx, row_ids, rows, F);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_sort_fields(const int64_t* offsets, const int64_t* sizes, const unsigned char* frozen, const int64_t* idx,
                                     int64_t* sorted_ids, int64_t* perm, int B, int F, int64_t max_vocab, void* stream) {
  FIL_CHECK_ARG(B >= 0 && F >= 1 && max_vocab >= 0);
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(offsets && idx && sorted_ids && perm);
  if (B > 8192) return fail(FIL_ERR_UNSUPPORTED, "fil_embed_sort_fields: B=%d > 8192 pairs per field (64 KiB of LDS); sort the row ids of fil_embed_row_ids instead", B);
  int N = 2, bits = 1;
  while (N < B) {
    N <<= 1;
    ++bits;
  }
  // 32-bit composites when every id + 1 (<= max_vocab, an upper bound of every field's vocabulary; 0 = unknown) fits above the position bits
  const bool narrow = max_vocab > 0 && ((max_vocab + 1) << bits) <= (int64_t)0xffffffffLL;
  const size_t sh = (size_t)N * (narrow ? sizeof(unsigned) : sizeof(unsigned long long));
  const void* kern = narrow ? reinterpret_cast<const void*>(embed_sort_fields_kernel<unsigned>)
                            : reinterpret_cast<const void*>(embed_sort_fields_kernel<unsigned long long>);
  if (sh > 48 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FIL_ERR_HIP, "fil_embed_sort_fields: cannot reserve %zu bytes of LDS", sh);
  }
  const dim3 block(N >= 2048 ? 1024 : std::max(64, N / 2));
  // (N = 1024 ... 8192: the keys in registers, embed_sort_fields_reg_kernel; FIL_EMBED_SORT_LDS=1 keeps the all-LDS network)
  static const bool lds_only = [] {
    const char* e = getenv("FIL_EMBED_SORT_LDS");
    return e != nullptr && e[0] == '1';
  }();
  const int E = N / 1024;
  if (!lds_only && N >= 1024 && N <= 8192) {
#define FIL_SORT_REG(KT, EV, PB)                                                                                                                     \
  hipLaunchKernelGGL((embed_sort_fields_reg_kernel<KT, EV>), dim3(F), dim3(1024), 0, (hipStream_t)stream, offsets, sizes, frozen, idx, sorted_ids, perm, \
                     B, F, PB)
    if (narrow) {
      if (E == 1) FIL_SORT_REG(unsigned, 1, bits);
      else if (E == 2) FIL_SORT_REG(unsigned, 2, bits);
      else if (E == 4) FIL_SORT_REG(unsigned, 4, bits);
      else FIL_SORT_REG(unsigned, 8, bits);
    } else {
      if (E == 1) FIL_SORT_REG(unsigned long long, 1, 32);
      else if (E == 2) FIL_SORT_REG(unsigned long long, 2, 32);
      else if (E == 4) FIL_SORT_REG(unsigned long long, 4, 32);
      else FIL_SORT_REG(unsigned long long, 8, 32);
    }
#undef FIL_SORT_REG
  } else if (narrow)
    hipLaunchKernelGGL(embed_sort_fields_kernel<unsigned>, dim3(F), block, sh, (hipStream_t)stream, offsets, sizes, frozen, idx, sorted_ids, perm, B, F, N, bits);
  else
    hipLaunchKernelGGL(embed_sort_fields_kernel<unsigned long long>, dim3(F), block, sh, (hipStream_t)stream, offsets, sizes, frozen, idx, sorted_ids, perm, B,
                       F, N, 32);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_embed_segment_sum(const float* g, const int64_t* perm, const int64_t* starts, const int64_t* rows_out, float* values,
                                     float* dtable, long U, int K, void* stream) {
  FIL_CHECK_ARG(U >= 0 && K >= 1);
  if (K > 256) return fail(FIL_ERR_UNSUPPORTED, "fil_embed_segment_sum: K=%d > 256", K);
  if (U == 0) return FIL_OK;
  FIL_CHECK_ARG(g && perm && starts && rows_out && (values || dtable));
  hipLaunchKernelGGL(embed_segment_sum_kernel, dim3((int)std::min<long>((U + 3) / 4, 256 * 16)), dim3(256), 0, (hipStream_t)stream, g, perm, starts,
                     rows_out, values, dtable, U, K);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
