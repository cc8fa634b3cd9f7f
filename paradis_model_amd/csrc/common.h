// Shared device helpers for libparadis_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/paradis_hip.h"

#define WAVE 64

void paradis_set_error(const char* fmt, ...);
// PARADIS_DETERMINISTIC=1 (read once): every reduction that would otherwise finish with float atomics
// runs in a fixed order (bit-identical gradients run to run), at a small cost; see DESIGN.md section 5b
bool paradis_deterministic();

#define PD_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      paradis_set_error(__VA_ARGS__);         \
      return 1;                               \
    }                                         \
  } while (0)

#define PD_CHECK_LAUNCH(name)                                               \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      paradis_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 2;                                                             \
    }                                                                       \
  } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property: a kernel family keeps one
// of these and asks `first()` before its launches (true once per device of the calling process).
struct PerDeviceOnce {
  unsigned long long mask[2] = {0ull, 0ull};   // up to 128 devices
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 128) return true;
    unsigned long long& m = mask[dev >> 6];
    const unsigned long long bit = 1ull << (dev & 63);
    if (m & bit) return false;
    m |= bit;
    return true;
  }
};

// ---------------------------------------------------------------------------
// Geocyclic index map (reference model/padding.py:11-39).
// (ii, jj) are IMAGE coordinates of a padded cell: ii in [-p, H+p), jj in [-p, W+p).
// Rows beyond a pole mirror about the pole row and shift by W/2; longitude wraps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int geo_wrap_col(int jj, int W) {
  return jj < 0 ? jj + W : (jj >= W ? jj - W : jj);
}

__device__ __forceinline__ void geo_src(int ii, int jj, int H, int W, int& r, int& c) {
  int j = geo_wrap_col(jj, W);
  if (ii < 0) {
    r = -ii;
    c = j + (W >> 1);
    if (c >= W) c -= W;
  } else if (ii >= H) {
    r = 2 * (H - 1) - ii;
    c = j + (W >> 1);
    if (c >= W) c -= W;
  } else {
    r = ii;
    c = j;
  }
}

// Enumerate every padded cell (image coords) that aliases source cell (y, x): f(ii, jj).
template <class F>
__device__ __forceinline__ void geo_for_each_alias(int y, int x, int H, int W, int p, F f) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int ii;
    bool mirror;
    if (t == 0) {
      ii = y;
      mirror = false;
    } else if (t == 1) {
      if (!(y >= 1 && y <= p)) continue;
      ii = -y;
      mirror = true;
    } else {
      if (!(y >= H - 1 - p && y <= H - 2)) continue;
      ii = 2 * (H - 1) - y;
      mirror = true;
    }
    int jb = x;
    if (mirror) {
      jb = x + (W >> 1);
      if (jb >= W) jb -= W;
    }
    f(ii, jb);
    if (jb >= W - p) f(ii, jb - W);
    if (jb < p) f(ii, jb + W);
  }
}

// unsigned maximum over the wave by DPP (no LDS-pipe shuffles, no waits): row_shr 1/2/4/8 leave each 16-lane
// row's maximum in its last lane, row_bcast 15 / 31 carry it on; LANE 63 ends up with the maximum of the wave.
// Lanes a shift has no source for read 0 (bound_ctrl): neutral for an unsigned maximum.
__device__ __forceinline__ uint32_t wave_umax_lane63(uint32_t v) {
#define PD_DPP_MAX(ctrl, rows) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, true))
  PD_DPP_MAX(0x111, 0xf); PD_DPP_MAX(0x112, 0xf); PD_DPP_MAX(0x114, 0xf); PD_DPP_MAX(0x118, 0xf);
  PD_DPP_MAX(0x142, 0xa); PD_DPP_MAX(0x143, 0xc);
#undef PD_DPP_MAX
  return v;
}
// Zero-fill by a KERNEL, not hipMemsetAsync: inside a captured HIP graph (harness.GraphedTrainStep) the memset
// nodes of ROCm 7.0 were observed to run out of order with their neighbouring kernel nodes when a replay starts on
// an idle device (NaN bias gradients: the atomics that accumulate into the buffer ran against stale contents);
// kernel nodes keep the stream order.
static __global__ void __launch_bounds__(256) pd_zero_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline hipError_t pd_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  const size_t n = (bytes + 3) / 4;       // every buffer zeroed here is a float / uint32 array
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(pd_zero_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

// Sum over the wave by DPP, the same value in every lane: row_shr 1/2/4/8 (lanes without a source add 0) leave each
// 16-lane row's sum in its last lane, row_bcast 15 / 31 carry it on to lane 63, one v_readlane broadcasts it.
// Seven dependent vector instructions instead of six LDS-pipe shuffles with a wait each (wave_sum below); the order of
// the additions differs from wave_sum's, the result agrees to rounding.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define PD_DPP_ADD(ctrl, rows) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xf, true))
  PD_DPP_ADD(0x111, 0xf); PD_DPP_ADD(0x112, 0xf); PD_DPP_ADD(0x114, 0xf); PD_DPP_ADD(0x118, 0xf);
  PD_DPP_ADD(0x142, 0xa); PD_DPP_ADD(0x143, 0xc);
#undef PD_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Whole geocyclic-padded plane [(H+2p) x (W+2p+xr)] into LDS, 16-byte path: the H x W interior is a plain copy (one float4
// load + two 8-byte LDS writes per four cells); only the halo columns (p on the left, p + xr on the right) and the 2p
// mirrored rows go through the index map.  ~4 loads and ~60 VALU instructions per thread instead of 10 and ~300.
// Needs W % 4 == 0, a 16-byte aligned plane, p and xr even (8-byte aligned LDS rows).
__device__ __forceinline__ void stage_plane_vec4(float* win, const float* __restrict__ F, int H, int W, int p, int xr = 0) {
  const int Wp = W + 2 * p + xr, Hp = H + 2 * p, tid = threadIdx.x;
  const int w4 = W >> 2, nvec = H * w4;
  const int hc = 2 * p + xr;
  const int nhalo_rows = 2 * p * Wp, nhalo = nhalo_rows + H * hc;
  constexpr int VB = 2, HB = 2;
  const float4* F4 = reinterpret_cast<const float4*>(F);
  for (int v0 = tid, k0 = tid; v0 < nvec || k0 < nhalo; v0 += 256 * VB, k0 += 256 * HB) {
    float4 q[VB];
    float hv[HB];
    int hdst[HB];
#pragma unroll
    for (int j = 0; j < VB; ++j) q[j] = F4[min(v0 + 256 * j, nvec - 1)];
#pragma unroll
    for (int j = 0; j < HB; ++j) {
      const int kk = k0 + 256 * j, k = min(kk, nhalo - 1);   // clamped: the load is unconditional
      int lr, lc;
      if (k < nhalo_rows) {          // the p rows beyond each pole, full padded width
        const int rr = k / Wp;
        lc = k - rr * Wp;
        lr = rr < p ? rr : Hp - 2 * p + rr;
      } else {                       // left / right halo columns of the interior rows
        const int e = k - nhalo_rows, rr = e / hc, cc = e - rr * hc;
        lr = rr + p;
        lc = cc < p ? cc : W + cc;
      }
      int sr, sc;
      geo_src(lr - p, lc - p, H, W, sr, sc);
      hv[j] = F[sr * W + sc];
      hdst[j] = kk < nhalo ? lr * Wp + lc : -1;
    }
#pragma unroll
    for (int j = 0; j < VB; ++j) {
      const int v = v0 + 256 * j;
      if (v < nvec) {
        const int y = v / w4, x4 = v - y * w4;
        float2* d = reinterpret_cast<float2*>(win + (y + p) * Wp + p + 4 * x4);   // 8-byte aligned (p, Wp even)
        d[0] = make_float2(q[j].x, q[j].y);
        d[1] = make_float2(q[j].z, q[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < HB; ++j)
      if (hdst[j] >= 0) win[hdst[j]] = hv[j];
  }
}


__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == PARADIS_ACT_SILU) {
    return z / (1.0f + expf(-z));
  } else if (act == PARADIS_ACT_GELU) {
    return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
  }
  return z;
}

__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == PARADIS_ACT_SILU) {
    float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
  } else if (act == PARADIS_ACT_GELU) {
    float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
    return cdf + z * pdf;
  }
  return 1.0f;
}
