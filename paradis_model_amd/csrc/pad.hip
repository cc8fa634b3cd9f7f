// a1: GeoCyclicPadding forward and adjoint (reference model/padding.py:11-39).
// Pure index arithmetic -> bit-exact.  HBM-bound: reads P, writes Pp floats per plane.
#include "common.h"

__global__ void __launch_bounds__(256)
geocyclic_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes,
                         int H, int W, int p) {
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  const int64_t per = (int64_t)Hp * Wp;
  const int64_t total = planes * per;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t plane = idx / per;
    int rem = (int)(idx - plane * per);
    int i = rem / Wp, j = rem - i * Wp;
    int r, c;
    geo_src(i - p, j - p, H, W, r, c);
    y[idx] = x[plane * (int64_t)H * W + (int64_t)r * W + c];
  }
}

__global__ void __launch_bounds__(256)
geocyclic_pad_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes,
                         int H, int W, int p) {
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  const int64_t per = (int64_t)H * W;
  const int64_t total = planes * per;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t plane = idx / per;
    int rem = (int)(idx - plane * per);
    int yy = rem / W, xx = rem - yy * W;
    const float* g = gy + plane * (int64_t)Hp * Wp;
    float acc = 0.f;
    geo_for_each_alias(yy, xx, H, W, p, [&](int ii, int jj) {
      acc += g[(int64_t)(ii + p) * Wp + (jj + p)];
    });
    gx[idx] = acc;
  }
}

static int check_pad(const char* name, int64_t planes, int H, int W, int p) {
  PD_REQUIRE(planes >= 0 && H >= 2 && W >= 2, "%s: bad shape", name);
  PD_REQUIRE(W % 2 == 0, "%s: Number of longitude points must be even", name);
  PD_REQUIRE(p >= 1 && p <= H - 2 && 2 * p <= W, "%s: pad width %d out of range for %dx%d", name, p, H, W);
  return 0;
}

extern "C" int paradis_geocyclic_pad_fwd(const float* x, float* y, int64_t planes, int H, int W,
                                         int p, void* stream) {
  if (int e = check_pad("geocyclic_pad_fwd", planes, H, W, p)) return e;
  if (planes == 0) return 0;
  int64_t total = planes * (int64_t)(H + 2 * p) * (W + 2 * p);
  int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 16);
  hipLaunchKernelGGL(geocyclic_pad_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y,
                     planes, H, W, p);
  PD_CHECK_LAUNCH("geocyclic_pad_fwd");
  return 0;
}

extern "C" int paradis_geocyclic_pad_bwd(const float* gy, float* gx, int64_t planes, int H, int W,
                                         int p, void* stream) {
  if (int e = check_pad("geocyclic_pad_bwd", planes, H, W, p)) return e;
  if (planes == 0) return 0;
  int64_t total = planes * (int64_t)H * W;
  int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 16);
  hipLaunchKernelGGL(geocyclic_pad_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gy, gx,
                     planes, H, W, p);
  PD_CHECK_LAUNCH("geocyclic_pad_bwd");
  return 0;
}
