// Rows f1-f3 of SURVEY.md section 8 (the callers right after / around the model in the training step):
//   f1  ParadisLoss forward + gradient in one pass   (reference utils/loss.py:233-282)
//   f2  rollout glue: strided channel-block copies that assemble the model input and the next
//       autoregressive input                          (reference trainer.py:534-538, 710-729)
//   f3  AdamW update, torch.optim.AdamW operation order (reference trainer.py:327-335)
// All HBM-bound streaming kernels.
#include <algorithm>
#include "common.h"

namespace {

// per element:  l = wf[c] * wl[h] * loss(pred - target);  grad = wf*wl*loss'(e) / N
// kind 0 = mse, 1 = smooth reversed Huber with threshold delta.  partial[block] = sum of l.
__global__ void __launch_bounds__(256)
paradis_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                    const float* __restrict__ wf, const float* __restrict__ wl,
                    float* __restrict__ grad, float* __restrict__ partial, int C, int H, int W,
                    int64_t total, int kind, float delta, float inv_n) {
  __shared__ float red[4];
  float acc = 0.f;
  const int HW = H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int hw = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int h = hw / W;
    const float wgt = wf[c] * (wl ? wl[h] : 1.0f);
    const float e = pred[i] - target[i];
    float l, dl;
    if (kind == 0) {
      l = e * e;
      dl = 2.0f * e;
    } else {
      const float a = fabsf(e);
      const float s = 1.0f / (1.0f + expf(-2.0f * (a - delta)));
      const float small = delta * a;
      const float large = (e * e + delta * delta) / (2.0f * delta);
      l = (1.0f - s) * small + s * large;
      const float ds = 2.0f * s * (1.0f - s);
      const float dla = ds * (large - small) + (1.0f - s) * delta + s * a / delta;
      dl = e > 0.f ? dla : (e < 0.f ? -dla : 0.f);
    }
    acc += l * wgt;
    if (grad) grad[i] = dl * wgt * inv_n;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
loss_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, float inv_n) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)inv_n);
}

__global__ void __launch_bounds__(256)
scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t n) {
  const float k = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = x[i] * k;
}

// dst[b, c, :] = src[b, c, :] for c < C: batch strides differ (channel-block copy)
__global__ void __launch_bounds__(256)
copy_channels_kernel(const float* __restrict__ src, int64_t src_bs, float* __restrict__ dst,
                     int64_t dst_bs, int B, int64_t per /* C*P */) {
  const int64_t total = (int64_t)B * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per, r = i - b * per;
    dst[b * dst_bs + r] = src[b * src_bs + r];
  }
}

// torch.optim.AdamW, single tensor, same operation order as the ATen implementation:
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g;
//   denom = sqrt(v)/sqrt(bc2) + eps;  p -= (lr/bc1) * m / denom
__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
             float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    float mi = m[i];
    mi = mi + (1.0f - b1) * (gi - mi);
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// One launch for a whole parameter group: block c works on chunk c = (tensor index, offset); all
// tensors share the hyper-parameters and the step count.  (335 per-tensor launches of ~5 us are
// launch-bound: 0.7 % of the training step.)
constexpr int ADAMW_CHUNK = 32768;
// dev (optional): {step count as float bits are NOT used - int32 step at [0], float lr at [1]} kept on the device so
// that a captured HIP graph of the training step stays valid from one replay to the next (the host-side step
// count and learning rate would be frozen into the kernel arguments): the bias corrections are then formed
// here, in double like the host path.
__global__ void __launch_bounds__(256)
adamw_multi_kernel(const int64_t* __restrict__ ptrs /* [4][T]: p, g, m, v */, const int64_t* __restrict__ numel,
                   const int* __restrict__ chunk_tensor, const int64_t* __restrict__ chunk_off, int T,
                   float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                   const int* __restrict__ dev) {
  if (dev) {     // (workgroup-uniform)
    // one thread forms the bias corrections (two double pow + a sqrt), the workgroup reads them from LDS.  Same formula
    // in double as the host path; the device's pow is not guaranteed to round like the host's libm, so the two paths
    // agree to ~1 ulp of the fp32 corrections, not bit for bit (tests/test_hip_train_rows.py checks 1e-6).
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
      const int step = dev[0];
      bc[0] = (float)(1.0 - pow((double)b1, (double)step));
      bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)step));
    }
    __syncthreads();
    lr = __int_as_float(dev[1]);
    bc1 = bc[0];
    bc2_sqrt = bc[1];
  }
  const int t = chunk_tensor[blockIdx.x];
  const int64_t off = chunk_off[blockIdx.x];
  const int64_t n = min((int64_t)ADAMW_CHUNK, numel[t] - off);
  float* p = reinterpret_cast<float*>(ptrs[t]) + off;
  const float* g = reinterpret_cast<const float*>(ptrs[T + t]) + off;
  float* m = reinterpret_cast<float*>(ptrs[2 * T + t]) + off;
  float* v = reinterpret_cast<float*>(ptrs[3 * T + t]) + off;
  const float step_size = lr / bc1;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    float mi = m[i];
    mi = mi + (1.0f - b1) * (gi - mi);
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ void adamw_tick_kernel(int* __restrict__ dev) { dev[0] += 1; }

inline int blocks_for(int64_t n) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 8));
}

}  // namespace

extern "C" int paradis_loss_blocks(int64_t total) { return blocks_for(total); }

extern "C" int paradis_loss_fwd_bwd(const float* pred, const float* target, const float* wf,
                                    const float* wl, float* loss, float* grad, float* partial, int B,
                                    int C, int H, int W, int kind, float delta, void* stream) {
  PD_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1, "loss_fwd_bwd: bad shape");
  PD_REQUIRE(kind == 0 || kind == 1, "loss_fwd_bwd: kind must be 0 (mse) or 1 (reversed_huber)");
  PD_REQUIRE(partial != nullptr && loss != nullptr, "loss_fwd_bwd: workspace/result missing");
  const int64_t total = (int64_t)B * C * H * W;
  const int nb = blocks_for(total);
  const float inv_n = (float)(1.0 / (double)total);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(paradis_loss_kernel, dim3(nb), dim3(256), 0, st, pred, target, wf, wl, grad, partial, C,
                     H, W, total, kind, delta, inv_n);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, partial, loss, nb, inv_n);
  PD_CHECK_LAUNCH("loss_fwd_bwd");
  return 0;
}

extern "C" int paradis_scale(const float* x, const float* scalar, float* y, int64_t n, void* stream) {
  PD_REQUIRE(n >= 0, "scale: bad size");
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, y, n);
  PD_CHECK_LAUNCH("scale");
  return 0;
}

extern "C" int paradis_copy_channels(const float* src, int64_t src_bs, float* dst, int64_t dst_bs, int B,
                                     int64_t per_sample, void* stream) {
  PD_REQUIRE(B >= 0 && per_sample >= 0, "copy_channels: bad size");
  if (B == 0 || per_sample == 0) return 0;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(blocks_for((int64_t)B * per_sample)), dim3(256), 0,
                     (hipStream_t)stream, src, src_bs, dst, dst_bs, B, per_sample);
  PD_CHECK_LAUNCH("copy_channels");
  return 0;
}

extern "C" int paradis_adamw_chunk(void) { return ADAMW_CHUNK; }

extern "C" int paradis_adamw_multi(const int64_t* ptrs, const int64_t* numel, const int* chunk_tensor,
                                   const int64_t* chunk_off, int n_tensors, int n_chunks, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step,
                                   const int* dev_state, void* stream) {
  PD_REQUIRE(n_tensors >= 0 && n_chunks >= 0 && (step >= 1 || dev_state != nullptr), "adamw_multi: bad arguments");
  if (n_chunks == 0) return 0;
  PD_REQUIRE(ptrs && numel && chunk_tensor && chunk_off, "adamw_multi: tables missing");
  const int st = step >= 1 ? step : 1;
  const float bc1 = (float)(1.0 - pow((double)beta1, st));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, ptrs, numel,
                     chunk_tensor, chunk_off, n_tensors, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, dev_state);
  PD_CHECK_LAUNCH("adamw_multi");
  return 0;
}

// dev_state[0] += 1 on the stream (the step count of a captured training step lives on the device)
extern "C" int paradis_adamw_tick(int* dev_state, void* stream) {
  PD_REQUIRE(dev_state != nullptr, "adamw_tick: state missing");
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dev_state);
  PD_CHECK_LAUNCH("adamw_tick");
  return 0;
}

extern "C" int paradis_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int step,
                                  void* stream) {
  PD_REQUIRE(n >= 0 && step >= 1, "adamw_step: bad arguments");
  if (n == 0) return 0;
  const float bc1 = (float)(1.0 - pow((double)beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
  PD_CHECK_LAUNCH("adamw_step");
  return 0;
}
