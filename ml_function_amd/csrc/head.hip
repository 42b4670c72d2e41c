// N2  the model's score head and loss, the handful of [B]-sized elementwise launches behind the interaction layers:
//   ScoreLayer(use_add=True).call  (core_layer/core_layer.py:58-84): keras Add over the [B,1] parts -> sigmoid
//   binary cross-entropy on probabilities as the reference compiles it (example/ctr_example/un_seq.py:61:
//   model.compile(loss=tf.losses.binary_crossentropy); TensorFlow 2.1 keras/backend.py binary_crossentropy: clip to
//   [eps, 1-eps], -(y log(p + eps) + (1-y) log(1 - p + eps)), mean over the batch)
// Each is ONE launch forward and one backward (torch: add, add, sigmoid | clamp, BCE, mean and eight launches of backward).
// Bound: launch latency (B floats); written for determinism -- the loss is summed by one workgroup in a fixed order.
#include "common.h"

namespace fil {

// out[i] = sigmoid(((a[i] + b[i]) + c[i]) + d[i]) -- the left-to-right order of keras Add / a chain of torch adds
__global__ __launch_bounds__(256) void score_add_sigmoid_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                    const float* __restrict__ c, const float* __restrict__ d,
                                                                    float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = a[i];
  if (b != nullptr) s += b[i];
  if (c != nullptr) s += c[i];
  if (d != nullptr) s += d[i];
  out[i] = 1.0f / (1.0f + expf(-s));
}

// dsum[i] = dout[i] * out[i] * (1 - out[i])   (the same gradient for every part of the sum)
__global__ __launch_bounds__(256) void score_add_sigmoid_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                                    float* __restrict__ dsum, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = out[i];
  dsum[i] = dout[i] * (p * (1.0f - p));
}

// loss = mean_i -(y log(pc + eps) + (1-y) log(1 - pc + eps)), pc = clip(p, eps, 1-eps);  dp[i] = d loss / d p[i] (0 where the clip
// is active, as the gradient of clip_by_value is).  One workgroup: thread t sums i = t, t + 1024, ... in order, the 1024 sums are folded by a
// fixed tree -- bit-identical repeats.
__global__ __launch_bounds__(1024) void bce_mean_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, float eps,
                                                            float* __restrict__ loss, float* __restrict__ dp, int n) {
  __shared__ float red[1024];
  const float inv_n = 1.0f / (float)n;
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float pi = p[i], yi = y[i];
    const float pc = fminf(fmaxf(pi, eps), 1.0f - eps);
    const float u = pc + eps, v = 1.0f - pc + eps;
    t -= yi * logf(u) + (1.0f - yi) * logf(v);
    if (dp != nullptr) {
      const bool inside = pi >= eps && pi <= 1.0f - eps;
      dp[i] = inside ? inv_n * ((1.0f - yi) / v - yi / u) : 0.f;
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_n;
}


// ---- MergeScoreLayer (core_layer/core_layer.py:86-100): StackLayer concat of the [B, w_i] parts -> Dense(O, softmax).  DeepFM / DCN end
// in it (model/models.py:87,104).  torch: cat, GEMM [B,D]x[D,O] (O = 2), bias add, softmax and their backward: ~12 launches for 4 KB
// of output.  Here ONE launch each way: the parts are read where they lie (no concatenated copy), fp32 or bf16 storage, fp32 math.
constexpr int kMergeMaxO = 8;       // Dense units of the head
constexpr int kMergeRows = 64;      // samples per workgroup of the backward (four waves x 16 rows)
struct MergeParts {
  const void* p[4];
  void* dp[4];
  int w[4];        // widths; 0 = unused
  int bf[4];       // 1: the part (and its gradient) is stored as bf16, 0: fp32
  int D;           // sum of the widths
};
__device__ __forceinline__ float merge_ld(const void* p, long i, int bf) {
  return bf ? (float)reinterpret_cast<const __hip_bfloat16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void merge_st(void* p, long i, float v, int bf) {
  if (bf) reinterpret_cast<__hip_bfloat16*>(p)[i] = (__hip_bfloat16)v;
  else reinterpret_cast<float*>(p)[i] = v;
}

// out[b, :] = softmax(concat_i parts_i[b, :] W + bias): a wave per sample (lanes over the D columns, O dot products folded by the wave)
template <int MO>   // MO >= O: 2 (the reference's output_dim = 2) or kMergeMaxO
__global__ __launch_bounds__(256) void merge_softmax_fwd_kernel(MergeParts mp, const float* __restrict__ W, const float* __restrict__ bias,
                                                                float* __restrict__ out, int B, int O) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float z[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) z[o] = 0.f;
  int d0 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = mp.w[i];
    for (int d = lane; d < w; d += 64) {
      const float x = merge_ld(mp.p[i], (long)b * w + d, mp.bf[i]);
      const float* wr = W + (long)(d0 + d) * O;
#pragma unroll
      for (int o = 0; o < MO; ++o)
        if (o < O) z[o] = fmaf(x, wr[o], z[o]);
    }
    d0 += w;
  }
#pragma unroll
  for (int o = 0; o < MO; ++o) z[o] = wave_sum(z[o]);
  if (lane == 0) {
    float m = -INFINITY;
    for (int o = 0; o < O; ++o) {
      z[o] += bias[o];
      m = fmaxf(m, z[o]);
    }
    float den = 0.f;
    for (int o = 0; o < O; ++o) {
      z[o] = expf(z[o] - m);
      den += z[o];
    }
    for (int o = 0; o < O; ++o) out[(long)b * O + o] = z[o] / den;
  }
}

// dz = out (dout - <dout, out>);  dparts_i[b, d] = sum_o dz[b, o] W[d, o];  block partials of dW[d, o] = sum_b x[b, d] dz[b, o] and
// db[o] = sum_b dz[b, o] over the workgroup's kMergeRows samples, summed in block order by block_partials_sum_kernel.  Wave w takes rows 16 w .. 16 w + 15 of the
// block, its lanes 64 columns at a time (16 loads in flight per lane); the four waves' sums of a column chunk meet in LDS in wave
// order.  (First form: 64 workgroups of 64 rows, a thread per column walking the rows one by one: 45 us for 2.4 MB.)
template <int MO>
__global__ __launch_bounds__(256) void merge_softmax_bwd_kernel(MergeParts mp, const float* __restrict__ W, const float* __restrict__ out,
                                                                const float* __restrict__ dout, float* __restrict__ part, int B, int O) {
  __shared__ float dz[kMergeRows][MO];
  __shared__ float red[4][64][MO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * kMergeRows, nrow = min(kMergeRows, B - b0), D = mp.D;
  // blockIdx.y = the 64-column chunk of the concatenated input this workgroup takes (round 6: one workgroup walked all chunks of
  // its rows one after the other -- three dependent rounds of loads behind the dz prologue on 64 CUs: 24.8 us for 2.4 MB)
  int pi = 0, dc = 0, d0 = 0;
  {
    int ch = blockIdx.y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nci = (mp.w[i] + 63) >> 6;
      if (pi == i && ch >= nci) {
        ch -= nci;
        d0 += mp.w[i];
        pi = i + 1;
      }
    }
    dc = ch * 64;
  }
  if (pi >= 4) return;
  const int w = mp.w[pi];
  const int d = dc + lane;
  const int r0 = wave * 16;     // this wave's rows of the block
  // the wave's sixteen rows of its column and the weights, requested before the dz prologue's loads are waited for
  float xs[16], wr[MO];
#pragma unroll
  for (int s = 0; s < 16; ++s) xs[s] = (d < w && r0 + s < nrow) ? merge_ld(mp.p[pi], (long)(b0 + r0 + s) * w + d, mp.bf[pi]) : 0.f;
#pragma unroll
  for (int o = 0; o < MO; ++o) wr[o] = (d < w && o < O) ? W[(long)(d0 + d) * O + o] : 0.f;
  for (int s = tid; s < kMergeRows; s += 256) {
    float dot = 0.f, p[MO], g[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) {
      p[o] = (s < nrow && o < O) ? out[(long)(b0 + s) * O + o] : 0.f;
      g[o] = (s < nrow && o < O) ? dout[(long)(b0 + s) * O + o] : 0.f;
      dot = fmaf(p[o], g[o], dot);
    }
#pragma unroll
    for (int o = 0; o < MO; ++o) dz[s][o] = p[o] * (g[o] - dot);
  }
  __syncthreads();
  float* mypart = part + (long)blockIdx.x * (D + 1) * O;
  float acc[MO];
#pragma unroll
  for (int o = 0; o < MO; ++o) acc[o] = 0.f;
  if (d < w) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int rr = r0 + s;
      float dx = 0.f;
#pragma unroll
      for (int o = 0; o < MO; ++o) {
        acc[o] = fmaf(xs[s], dz[rr][o], acc[o]);
        dx = fmaf(dz[rr][o], wr[o], dx);
      }
      if (rr < nrow && mp.dp[pi] != nullptr) merge_st(mp.dp[pi], (long)(b0 + rr) * w + d, dx, mp.bf[pi]);
    }
  }
#pragma unroll
  for (int o = 0; o < MO; ++o) red[wave][lane][o] = acc[o];
  __syncthreads();
  if (wave == 0 && d < w) {
    for (int o = 0; o < O; ++o) mypart[(long)(d0 + d) * O + o] = (red[0][lane][o] + red[1][lane][o]) + (red[2][lane][o] + red[3][lane][o]);
  }
  if (blockIdx.y == 0 && tid < O) {
    float t = 0.f;
    for (int s = 0; s < nrow; ++s) t += dz[s][tid];
    mypart[(long)D * O + tid] = t;
  }
}

// out[i] (i < n1) | out2[i - n1] = sum over the nblk block partials part[blk][n] in block order, eight in flight (a second, tiny launch:
// a ticketed in-kernel reduction by the last workgroup needs an agent-scope release per workgroup, which on this chip writes the whole
// L2 back -- 45-55 us behind kernels that left it dirty, measured -- where the launch boundary costs ~3 us)
__global__ __launch_bounds__(256) void block_partials_sum_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out, int n1,
                                                                 float* __restrict__ out2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int blk = 0;
  for (; blk + 7 < nblk; blk += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += part[(long)(blk + u) * n + i];
  }
  for (; blk < nblk; ++blk) t[0] += part[(long)blk * n + i];
  const float v = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  if (i < n1) out[i] = v;
  else out2[i - n1] = v;
}

// ---- Dense + bias + ReLU of the zoo's MLPs (DnnLayer: core_layer/core_layer.py:102-118,201-226), backward half.  The two GEMMs of a
// layer's backward stay library GEMMs; what torch runs around them -- threshold_backward over [B, N], then a column
// reduce of the same [B, N] for the bias gradient -- is ONE pass here: dz = dy where y > 0 (else 0), written in dy's storage type, and
// dbias[n] = sum_b dz[b, n] through block partials summed in block order by block_partials_sum_kernel (deterministic).
constexpr int kReluRows = 64;   // rows per workgroup
__device__ __forceinline__ void relu_ld4(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void relu_ld4(const __hip_bfloat16* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
}
__device__ __forceinline__ void relu_st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void relu_st4(__hip_bfloat16* p, const float (&v)[4]) {   // (the values are dy's own bf16 values or zero: exact)
  uint2 t;
  t.x = (__builtin_bit_cast(unsigned, v[0]) >> 16) | (__builtin_bit_cast(unsigned, v[1]) & 0xffff0000u);
  t.y = (__builtin_bit_cast(unsigned, v[2]) >> 16) | (__builtin_bit_cast(unsigned, v[3]) & 0xffff0000u);
  *reinterpret_cast<uint2*>(p) = t;
}
// VEC (N % 4 == 0, N <= 1024): a thread owns four consecutive columns and every RL-th row of the block's 32 (16- / 8-byte accesses,
// four rows in flight); the row lanes' sums meet through LDS in lane order.  Otherwise: a thread per column, row by row.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void relu_bias_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dz, float* __restrict__ part,
                                                            int B, int N) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, b0 = blockIdx.x * kReluRows, nrow = min(kReluRows, B - b0);
  if constexpr (VEC) {
    const int Q = N >> 2, RL = 256 / Q, q = tid % Q, rl = tid / Q;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
      for (int r0 = rl; r0 < nrow; r0 += 4 * RL) {
        float yv[4][4], gv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * RL;
          if (r < nrow) {
            relu_ld4(y + (long)(b0 + r) * N + 4 * q, yv[u]);
            relu_ld4(dy + (long)(b0 + r) * N + 4 * q, gv[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * RL;
          if (r < nrow) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              gv[u][e] = yv[u][e] > 0.f ? gv[u][e] : 0.f;
              acc[e] += gv[u][e];
            }
            relu_st4(dz + (long)(b0 + r) * N + 4 * q, gv[u]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) red[rl * N + 4 * q + e] = acc[e];
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
      float t = 0.f;
      for (int l = 0; l < RL; ++l) t += red[l * N + n];
      part[(long)blockIdx.x * N + n] = t;
    }
  } else {
    for (int n = tid; n < N; n += 256) {
      float t = 0.f;
      for (int s = 0; s < nrow; ++s) {
        const long i = (long)(b0 + s) * N + n;
        const float g = (float)y[i] > 0.f ? (float)dy[i] : 0.f;
        dz[i] = (T)g;
        t += g;
      }
      part[(long)blockIdx.x * N + n] = t;
    }
  }
}

}  // namespace fil

using namespace fil;

extern "C" int fil_score_add_sigmoid_fwd(const float* a, const float* b, const float* c, const float* d, float* out, int n,
                                         void* stream) {
  FIL_CHECK_ARG(n >= 0);
  if (n == 0) return FIL_OK;
  FIL_CHECK_ARG(a != nullptr && out != nullptr);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("score_fwd", st, 4.0 * n * (2 + (b != nullptr) + (c != nullptr) + (d != nullptr)));
  hipLaunchKernelGGL(score_add_sigmoid_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a, b, c, d, out, n);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_score_add_sigmoid_bwd(const float* out, const float* dout, float* dsum, int n, void* stream) {
  FIL_CHECK_ARG(n >= 0);
  if (n == 0) return FIL_OK;
  FIL_CHECK_ARG(out != nullptr && dout != nullptr && dsum != nullptr);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("score_bwd", st, 12.0 * n);
  hipLaunchKernelGGL(score_add_sigmoid_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, out, dout, dsum, n);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_bce_mean_fwd(const float* p, const float* y, float eps, float* loss, float* dp, int n, void* stream) {
  FIL_CHECK_ARG(n >= 1);
  FIL_CHECK_ARG(eps >= 0.f && eps < 0.5f);
  FIL_CHECK_ARG(p != nullptr && y != nullptr && loss != nullptr);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("bce_fwd", st, 4.0 * n * (dp != nullptr ? 3 : 2));
  hipLaunchKernelGGL(bce_mean_fwd_kernel, dim3(1), dim3(1024), 0, st, p, y, eps, loss, dp, n);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

static int merge_check(const char* fn, const void* const* parts, const int* widths, const int* dtypes, int n_parts, int B, int O, MergeParts& mp) {
  if (B < 0 || n_parts < 1 || n_parts > 4 || O < 1 || O > kMergeMaxO || parts == nullptr || widths == nullptr || dtypes == nullptr)
    return fail(FIL_ERR_ARG, "%s: B=%d, %d parts (1..4), %d units (1..%d)", fn, B, n_parts, O, kMergeMaxO);
  mp.D = 0;
  for (int i = 0; i < 4; ++i) {
    mp.p[i] = i < n_parts ? parts[i] : nullptr;
    mp.dp[i] = nullptr;
    mp.w[i] = i < n_parts ? widths[i] : 0;
    mp.bf[i] = 0;
    if (i < n_parts) {
      if (dtypes[i] != FIL_F32 && dtypes[i] != FIL_BF16) return fail(FIL_ERR_ARG, "%s: part %d: dtype %d (f32 or bf16 storage)", fn, i, dtypes[i]);
      mp.bf[i] = dtypes[i] == FIL_BF16;
    }
    if (i < n_parts && (widths[i] < 1 || (B > 0 && parts[i] == nullptr))) return fail(FIL_ERR_ARG, "%s: part %d: width %d", fn, i, widths[i]);
    mp.D += mp.w[i];
  }
  if (mp.D > 8192) return fail(FIL_ERR_UNSUPPORTED, "%s: %d concatenated columns > 8192", fn, mp.D);
  return FIL_OK;
}

extern "C" size_t fil_merge_softmax_bwd_workspace_bytes(int B, int D, int O) {
  return 256 + (size_t)cdiv(std::max(B, 1), kMergeRows) * (size_t)(D + 1) * O * sizeof(float);
}

extern "C" int fil_merge_softmax_fwd(const void* const* parts, const int* widths, const int* dtypes, int n_parts, const float* W, const float* bias,
                                     float* out, int B, int O, void* stream) {
  MergeParts mp;
  const int rc = merge_check("fil_merge_softmax_fwd", parts, widths, dtypes, n_parts, B, O, mp);
  if (rc != FIL_OK) return rc;
  if (B == 0) return FIL_OK;
  FIL_CHECK_ARG(W != nullptr && bias != nullptr && out != nullptr);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("merge_softmax_fwd", st, (double)B * (mp.D * 4.0 + O * 4.0));
  if (O <= 2) hipLaunchKernelGGL(merge_softmax_fwd_kernel<2>, dim3(cdiv(B, 4)), dim3(256), 0, st, mp, W, bias, out, B, O);
  else hipLaunchKernelGGL(merge_softmax_fwd_kernel<kMergeMaxO>, dim3(cdiv(B, 4)), dim3(256), 0, st, mp, W, bias, out, B, O);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" int fil_merge_softmax_bwd(const void* const* parts, const int* widths, const int* dtypes, int n_parts, const float* W, const float* out,
                                     const float* dout, void* const* dparts, float* dW, float* db, int B, int O, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  MergeParts mp;
  const int rc = merge_check("fil_merge_softmax_bwd", parts, widths, dtypes, n_parts, B, O, mp);
  if (rc != FIL_OK) return rc;
  FIL_CHECK_ARG(dW != nullptr && db != nullptr);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    (void)hipMemsetAsync(dW, 0, (size_t)mp.D * O * sizeof(float), st);
    (void)hipMemsetAsync(db, 0, (size_t)O * sizeof(float), st);
    return FIL_OK;
  }
  FIL_CHECK_ARG(W != nullptr && out != nullptr && dout != nullptr);
  if (workspace == nullptr || workspace_bytes < fil_merge_softmax_bwd_workspace_bytes(B, mp.D, O))
    return fail(FIL_ERR_WORKSPACE, "fil_merge_softmax_bwd: workspace %zu < %zu bytes", workspace_bytes, fil_merge_softmax_bwd_workspace_bytes(B, mp.D, O));
  for (int i = 0; i < n_parts; ++i) mp.dp[i] = dparts != nullptr ? dparts[i] : nullptr;
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);   // block partials [nblk][(D + 1) O]
  ProfScope ps("merge_softmax_bwd", st, (double)B * (2.0 * mp.D * 4.0 + 2.0 * O * 4.0));
  const int nblk = cdiv(B, kMergeRows), n = (mp.D + 1) * O;
  int nch = 0;   // 64-column chunks of the parts: a workgroup per (block of rows, chunk)
  for (int i = 0; i < 4; ++i) nch += (mp.w[i] + 63) / 64;
  if (O <= 2) hipLaunchKernelGGL(merge_softmax_bwd_kernel<2>, dim3(nblk, nch), dim3(256), 0, st, mp, W, out, dout, part, B, O);
  else hipLaunchKernelGGL(merge_softmax_bwd_kernel<kMergeMaxO>, dim3(nblk, nch), dim3(256), 0, st, mp, W, out, dout, part, B, O);
  hipLaunchKernelGGL(block_partials_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, part, nblk, n, dW, mp.D * O, db);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}

extern "C" size_t fil_relu_bias_bwd_workspace_bytes(int B, int N) { return 256 + (size_t)cdiv(std::max(B, 1), kReluRows) * (size_t)N * sizeof(float); }

extern "C" int fil_relu_bias_bwd(const void* y, const void* dy, void* dz, float* dbias, int B, int N, int dtype, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  FIL_CHECK_ARG(B >= 0 && N >= 1 && dbias != nullptr);
  if (dtype != FIL_F32 && dtype != FIL_BF16) return fail(FIL_ERR_ARG, "fil_relu_bias_bwd: dtype %d (f32 or bf16 storage)", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    (void)hipMemsetAsync(dbias, 0, (size_t)N * sizeof(float), st);
    return FIL_OK;
  }
  FIL_CHECK_ARG(y != nullptr && dy != nullptr && dz != nullptr);
  if (workspace == nullptr || workspace_bytes < fil_relu_bias_bwd_workspace_bytes(B, N))
    return fail(FIL_ERR_WORKSPACE, "fil_relu_bias_bwd: workspace %zu < %zu bytes", workspace_bytes, fil_relu_bias_bwd_workspace_bytes(B, N));
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);   // block partials [nblk][N]
  ProfScope ps("relu_bias_bwd", st, (double)B * N * 3.0 * (dtype == FIL_F32 ? 4.0 : 2.0));
  const int nblk = cdiv(B, kReluRows);
  const bool vec = (N & 3) == 0 && N <= 1024;
#define FIL_RELU(T, V) \
  hipLaunchKernelGGL((relu_bias_bwd_kernel<T, V>), dim3(nblk), dim3(256), 0, st, (const T*)y, (const T*)dy, (T*)dz, part, B, N)
  if (dtype == FIL_F32) {
    if (vec) FIL_RELU(float, true); else FIL_RELU(float, false);
  } else {
    if (vec) FIL_RELU(__hip_bfloat16, true); else FIL_RELU(__hip_bfloat16, false);
  }
#undef FIL_RELU
  hipLaunchKernelGGL(block_partials_sum_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, part, nblk, N, dbias, N, (float*)nullptr);
  FIL_CHECK_LAUNCH();
  return FIL_OK;
}
