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
