// a8: ChannelNorm (reference model/blocks.py:118-134): per-pixel normalisation over channels with
// the unbiased variance, y = (x - mean) * (var + eps)^-1/2 * w[c] + b[c].
// Layout [B, C, P]: the reduction runs over C with stride P, so lanes are laid along the pixel
// axis (coalesced 256-B rows) and waves split the channel axis.  HBM-bound: 8 B per (c, pixel)
// forward.  The input may be the virtual concatenation of two tensors (reference
// model/paradis.py:249, cat([hidden, hidden_static])) so the cat is never materialised.
#include <algorithm>
#include "common.h"

namespace {

constexpr int NPX = 64;  // pixels per workgroup (one wave width)

struct CatSrc {
  const float* x1; const float* x2;
  int C1, C2;
  int64_t bs1, bs2;
  __device__ __forceinline__ const float* row(int b, int c, int P) const {
    return c < C1 ? x1 + (int64_t)b * bs1 + (int64_t)c * P : x2 + (int64_t)b * bs2 + (int64_t)(c - C1) * P;
  }
};

// 1024 threads = NPXF pixels x G = 1024/NPXF channel groups; group g owns channels c = g + G*i.
// MAXV values per thread stay in registers so x is read from HBM exactly once (MAXV = 0: three
// passes, any C).  NPXF = 32 (128-byte row segments, 36 values per thread, ~64 VGPRs) keeps two
// workgroups per CU so that one loads while the other stores; with 64 pixels x 72 values only one
// 1024-thread workgroup fits and its load / reduce / store phases run back to back (2.7 TB/s).
// Y16 (round 6, bf16-mixed mode): y is written as bf16 (round to nearest even) for a consumer that is a pointwise GEMM -
// the value that GEMM would round its operand to anyway (the reference casts the norm's output to bf16 at the conv2d,
// model/blocks.py:86 under train.py:56), in half the bytes.
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float bf16_widen(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

template <int MAXV, int NPXF, bool Y16 = false>
__global__ void __launch_bounds__(1024)
channel_norm_fwd_kernel(CatSrc s, const float* __restrict__ w, const float* __restrict__ bias,
                        float* __restrict__ y, float* __restrict__ mean_out,
                        float* __restrict__ rstd_out, int P, int tiles, float eps) {
  constexpr int G = 1024 / NPXF;
  __shared__ float red[G][NPXF];
  __shared__ float stat[2][NPXF];
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NPXF;
  const int lane = threadIdx.x % NPXF, grp = threadIdx.x / NPXF;
  const int p = p0 + lane;
  const bool live = p < P;
  float vals[MAXV > 0 ? MAXV : 1];

  float sum = 0.f;
  if (MAXV > 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = grp + G * i;
      vals[i] = (live && c < C) ? s.row(b, c, P)[p] : 0.f;
      sum += vals[i];
    }
  } else {
    for (int c = grp; c < C; c += G) sum += live ? s.row(b, c, P)[p] : 0.f;
  }
  red[grp][lane] = sum;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) t += red[k][lane];
    stat[0][lane] = t / (float)C;
  }
  __syncthreads();
  const float mean = stat[0][lane];
  float sq = 0.f;
  if (MAXV > 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = grp + G * i;
      const float d = (c < C) ? vals[i] - mean : 0.f;
      sq += d * d;
    }
  } else {
    for (int c = grp; c < C; c += G) {
      const float d = live ? s.row(b, c, P)[p] - mean : 0.f;
      sq += d * d;
    }
  }
  __syncthreads();
  red[grp][lane] = sq;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) t += red[k][lane];
    const float var = t / (float)(C - 1);
    const float r = 1.0f / sqrtf(var + eps);
    stat[1][lane] = r;
    if (live) {
      mean_out[(int64_t)b * P + p] = mean;
      rstd_out[(int64_t)b * P + p] = r;
    }
  }
  __syncthreads();
  const float rstd = stat[1][lane];
  if (live) {
    float* yb = y + (int64_t)b * C * P + p;
    uint16_t* yb16 = reinterpret_cast<uint16_t*>(y) + (int64_t)b * C * P + p;
    if (MAXV > 0) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = grp + G * i;
        if (c < C) {
          const float v = (vals[i] - mean) * rstd * w[c] + bias[c];
          if constexpr (Y16) yb16[(int64_t)c * P] = bf16_bits(v); else yb[(int64_t)c * P] = v;
        }
      }
    } else {
      for (int c = grp; c < C; c += G) {
        const float v = (s.row(b, c, P)[p] - mean) * rstd * w[c] + bias[c];
        if constexpr (Y16) yb16[(int64_t)c * P] = bf16_bits(v); else yb[(int64_t)c * P] = v;
      }
    }
  }
}

// The same 32-pixel x 32-group kernel for even C1, C2 with the register diet that two 1024-thread workgroups per CU
// need (<= 64 VGPRs; the generic kernel above compiles to 102: one 64-bit address per load, and the affine parameters
// of all MAXV channels hoisted in front of the stores).  A wave holds two channel groups (lanes 0-31: channel
// 2 wave + 32 i, lanes 32-63: the next one), so the row pair of value i has a wave-uniform base: every load and store
// is `scalar base + one 32-bit lane offset`; w and bias are staged in LDS once and read at the point of use.
template <int MAXV, bool Y16 = false>
__global__ void __launch_bounds__(1024, 8)
channel_norm_fwd32_kernel(CatSrc s, const float* __restrict__ w, const float* __restrict__ bias,
                          float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                          int P, int tiles, float eps) {
  constexpr int G = 32, NPXF = 32;
  __shared__ float red[G][NPXF];
  __shared__ float stat[2][NPXF];
  __shared__ float wl[G * MAXV], bl[G * MAXV];
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NPXF;
  const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = p0 + lane;
  const bool live = p < P;
  for (int c = tid; c < C; c += 1024) { wl[c] = w[c]; bl[c] = bias[c]; }
  const unsigned voff = (unsigned)(((tid >> 5) & 1) * P + min(p, P - 1));   // (second group of the wave: one row further)
  typedef const __attribute__((address_space(1))) float* gptr;
  auto rowbase = [&](int ce) __attribute__((always_inline)) -> gptr {   // wave-uniform: channel ce (even) of sample b
    const float* r = s.row(b, ce, P);
    const uint64_t a = (uint64_t)r;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (gptr)(((uint64_t)hi << 32) | lo);
  };
  float vals[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ce = 2 * wave + G * i;                   // both groups of the wave are valid or not together (C even)
    vals[i] = ce < C ? rowbase(ce)[voff] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) sum += vals[i];
  red[grp][lane] = sum;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) t += red[k][lane];
    stat[0][lane] = t / (float)C;
  }
  __syncthreads();
  const float mean = stat[0][lane];
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d = (2 * wave + G * i < C) ? vals[i] - mean : 0.f;
    sq += d * d;
  }
  __syncthreads();
  red[grp][lane] = sq;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) t += red[k][lane];
    const float r = 1.0f / sqrtf(t / (float)(C - 1) + eps);
    stat[1][lane] = r;
    if (live) {
      mean_out[(int64_t)b * P + p] = mean;
      rstd_out[(int64_t)b * P + p] = r;
    }
  }
  __syncthreads();
  const float rstd = stat[1][lane];
  if (!live) return;
  typedef __attribute__((address_space(1))) float* gwptr;
  typedef __attribute__((address_space(1))) uint16_t* gwptr16;
  constexpr int ES = Y16 ? 2 : 4;
  const uint64_t ybase = (uint64_t)y + (uint64_t)b * C * P * ES;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ce = 2 * wave + G * i;
    if (ce < C) {
      const int c = grp + G * i;
      const float v = (vals[i] - mean) * rstd * wl[c] + bl[c];
      const uint64_t a = ybase + (uint64_t)ce * P * ES;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
      if constexpr (Y16) ((gwptr16)(((uint64_t)hi << 32) | lo))[voff] = bf16_bits(v);
      else ((gwptr)(((uint64_t)hi << 32) | lo))[voff] = v;
    }
  }
}

// gx = rstd * ( g*w - mean_c(g*w) - xhat * sum_c(g*w*xhat)/(C-1) )
__global__ void __launch_bounds__(1024)
channel_norm_bwd_dx_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ w,
                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                           float* __restrict__ gx1, float* __restrict__ gx2, int64_t gbs1,
                           int64_t gbs2, const float* __restrict__ add1, int64_t abs1, int P, int tiles) {
  __shared__ float red[2][16][NPX];
  __shared__ float stat[2][NPX];
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NPX;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = p0 + lane;
  const bool live = p < P;
  const float mean = live ? mean_in[(int64_t)b * P + p] : 0.f;
  const float rstd = live ? rstd_in[(int64_t)b * P + p] : 0.f;
  const float* gyb = gy + (int64_t)b * C * P + p;
  float s1 = 0.f, s2 = 0.f;
  for (int c = grp; c < C; c += 16) {
    if (live) {
      const float gh = gyb[(int64_t)c * P] * w[c];
      const float xh = (s.row(b, c, P)[p] - mean) * rstd;
      s1 += gh;
      s2 += gh * xh;
    }
  }
  red[0][grp][lane] = s1;
  red[1][grp][lane] = s2;
  __syncthreads();
  if (grp < 2) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[grp][k][lane];
    stat[grp][lane] = t;
  }
  __syncthreads();
  if (live) {
    const float m1 = stat[0][lane] / (float)C, m2 = stat[1][lane] / (float)(C - 1);
    for (int c = grp; c < C; c += 16) {
      const float gh = gyb[(int64_t)c * P] * w[c];
      const float xh = (s.row(b, c, P)[p] - mean) * rstd;
      const float v = rstd * (gh - m1 - xh * m2);
      if (c < s.C1) {
        const float o = v + (add1 ? add1[(int64_t)b * abs1 + (int64_t)c * P + p] : 0.f);
        gx1[(int64_t)b * gbs1 + (int64_t)c * P + p] = o;
      } else if (gx2) gx2[(int64_t)b * gbs2 + (int64_t)(c - s.C1) * P + p] = v;
    }
  }
}

// gw[c] = sum_{b,p} gy * xhat ; gb[c] = sum_{b,p} gy.  grid (C, chunks): partial[c][chunk][2]
__global__ void __launch_bounds__(256)
channel_norm_bwd_dw_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ mean_in,
                           const float* __restrict__ rstd_in, float* __restrict__ partial, int B,
                           int P, int chunks) {
  __shared__ float red[2][4];
  const int C = s.C1 + s.C2;
  const int c = blockIdx.x / chunks, chunk = blockIdx.x - c * chunks;
  const int64_t total = (int64_t)B * P;
  float aw = 0.f, ab = 0.f;
  for (int64_t i = (int64_t)chunk * 256 + threadIdx.x; i < total; i += (int64_t)chunks * 256) {
    const int b = (int)(i / P), p = (int)(i - (int64_t)b * P);
    const float g = gy[((int64_t)b * C + c) * P + p];
    const float xh = (s.row(b, c, P)[p] - mean_in[i]) * rstd_in[i];
    aw += g * xh;
    ab += g;
  }
  aw = wave_sum_dpp(aw);
  ab = wave_sum_dpp(ab);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = aw; red[1][wave] = ab; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    partial[((int64_t)c * chunks + chunk) * 2 + threadIdx.x] = t;
  }
}

__global__ void __launch_bounds__(256)
channel_norm_bwd_finish(const float* __restrict__ partial, float* __restrict__ gw,
                        float* __restrict__ gb, int C, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < chunks; ++k) {
    a += partial[((int64_t)c * chunks + k) * 2];
    b += partial[((int64_t)c * chunks + k) * 2 + 1];
  }
  gw[c] = a;
  gb[c] = b;
}

// Fused backward: one pass over (gy, x) produces the input gradient AND the per-channel partial sums
// of the weight/bias gradients.  1024 threads = 32 pixels x 32 channel groups; a thread keeps its
// <= MAXC gy values in registers and parks xhat in LDS (32 px x C floats <= 147 KiB of the 160 KiB),
// so gy and x are read from HBM exactly once (the three-kernel path reads each twice and then once
// more for the parameter gradients).
// partial layout: [block][2][C]  (row 0: sum gy*xhat, row 1: sum gy) over the block's 32 pixels.
template <int MAXC, int NPB>   // NPB pixels x 32 channel groups per workgroup (NPB = 32 or 16)
__global__ void __launch_bounds__(NPB * 32)
channel_norm_bwd_fused_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ w,
                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                              float* __restrict__ gx1, float* __restrict__ gx2, int64_t gbs1,
                              int64_t gbs2, const float* __restrict__ add1, int64_t abs1, float* __restrict__ partial, int P, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][32][NPB] reduce + [C][NPB] xhat
  float (*red)[32][NPB] = reinterpret_cast<float (*)[32][NPB]>(lds);
  float* xs = lds + 2 * 32 * NPB;
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NPB;
  const int px = threadIdx.x % NPB, grp = threadIdx.x / NPB;
  const int p = p0 + px;
  const bool live = p < P;
  const float mean = live ? mean_in[(int64_t)b * P + p] : 0.f;
  const float rstd = live ? rstd_in[(int64_t)b * P + p] : 0.f;
  float g[MAXC];
  float s1 = 0.f, s2 = 0.f;
  // loads in batches from clamped (always valid) addresses, selected afterwards: a load under a
  // per-lane condition is waited for on its own
  const int pc = min(p, P - 1);
  const float* gyc = gy + (int64_t)b * C * P + pc;
  constexpr int CH = MAXC % 6 == 0 ? 6 : 4;
#pragma unroll
  for (int i0 = 0; i0 < MAXC; i0 += CH) {
    float xv[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = min(grp + 32 * (i0 + j), C - 1);
      g[i0 + j] = gyc[(int64_t)c * P];
      xv[j] = s.row(b, c, P)[pc];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = grp + 32 * (i0 + j);
      const bool on = live && c < C;
      g[i0 + j] = on ? g[i0 + j] : 0.f;
      if (c < C) {
        const float xh = on ? (xv[j] - mean) * rstd : 0.f;
        xs[c * NPB + px] = xh;
        const float gh = g[i0 + j] * w[c];
        s1 += gh;
        s2 += gh * xh;
      }
    }
  }
  red[0][grp][px] = s1;
  red[1][grp][px] = s2;
  __syncthreads();
  if (grp < 2) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[grp][k][px];
    red[grp][0][px] = t;
  }
  __syncthreads();
  const float m1 = red[0][0][px] / (float)C, m2 = red[1][0][px] / (float)(C - 1);
  float* pw = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = grp + 32 * i;
    if (c < C) {
      const float xh = xs[c * NPB + px];
      if (live) {
        const float v = rstd * (g[i] * w[c] - m1 - xh * m2);
        if (c < s.C1) {
          const float o = v + (add1 ? add1[(int64_t)b * abs1 + (int64_t)c * P + p] : 0.f);
          gx1[(int64_t)b * gbs1 + (int64_t)c * P + p] = o;
        } else if (gx2) gx2[(int64_t)b * gbs2 + (int64_t)(c - s.C1) * P + p] = v;
      }
      float a = g[i] * xh, d = g[i];       // dead pixels hold zeros
#pragma unroll
      for (int o = NPB / 2; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, NPB);
        d += __shfl_xor(d, o, NPB);
      }
      if (px == 0) { pw[c] = a; pw[C + c] = d; }
    }
  }
}

// Same work with nothing parked between the two phases: gy and x are streamed twice, the second time
// a few microseconds after the first (the tile is 32 px x C x 8 B = 295 KiB: served by L2 / the
// Infinity Cache, HBM traffic unchanged).  512 threads = 32 pixels x 16 channel groups, 4 KiB of LDS,
// ~40 VGPRs: four workgroups share a CU, so loads, reductions and stores of different tiles overlap
// (the register/LDS-resident version fits one workgroup per CU and runs its phases back to back).
template <bool HAS_ADD>
__global__ void __launch_bounds__(512)
channel_norm_bwd_reread_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ w,
                               const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                               float* __restrict__ gx1, float* __restrict__ gx2, int64_t gbs1,
                               int64_t gbs2, const float* __restrict__ add1, int64_t abs1, float* __restrict__ partial, int P, int tiles) {
  constexpr int NPB = 32, G = 16;
  __shared__ float red[2][G][NPB];
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NPB;
  const int px = threadIdx.x % NPB, grp = threadIdx.x / NPB;
  const int p = min(p0 + px, P - 1);          // clamped: loads are unconditional, stores predicated
  const bool live = p0 + px < P;
  const float mean = mean_in[(int64_t)b * P + p];
  const float rstd = rstd_in[(int64_t)b * P + p];
  const float* gyb = gy + (int64_t)b * C * P + p;
  float s1 = 0.f, s2 = 0.f;
  constexpr int U = 6;   // channels per batch: 2*U loads in flight per thread, issued before their use
  for (int c0 = grp; c0 < C; c0 += G * U) {
    float gv[U], xv[U], wv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int c = min(c0 + G * j, C - 1);
      gv[j] = gyb[(int64_t)c * P];
      xv[j] = s.row(b, c, P)[p];
      wv[j] = c0 + G * j < C ? w[c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const float gh = gv[j] * wv[j];
      s1 += gh;
      s2 += gh * ((xv[j] - mean) * rstd);
    }
  }
  red[0][grp][px] = live ? s1 : 0.f;
  red[1][grp][px] = live ? s2 : 0.f;
  __syncthreads();
  if (grp < 2) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) t += red[grp][k][px];
    red[grp][0][px] = t;
  }
  __syncthreads();
  const float m1 = red[0][0][px] / (float)C, m2 = red[1][0][px] / (float)(C - 1);
  float* pw = partial + (int64_t)blockIdx.x * 2 * C;
  for (int c0 = grp; c0 < C; c0 += G * U) {
    float gv[U], xv[U], wv[U], av[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int c = min(c0 + G * j, C - 1);
      gv[j] = gyb[(int64_t)c * P];
      xv[j] = s.row(b, c, P)[p];
      wv[j] = w[c];
      av[j] = HAS_ADD ? add1[(int64_t)b * abs1 + (int64_t)min(c, s.C1 - 1) * P + p] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int c = c0 + G * j;
      if (c < C) {
        const float g = live ? gv[j] : 0.f;     // dead pixels contribute zeros
        const float xh = (xv[j] - mean) * rstd;
        if (live) {
          const float v = rstd * (g * wv[j] - m1 - xh * m2);
          if (c < s.C1) {
            gx1[(int64_t)b * gbs1 + (int64_t)c * P + p] = v + av[j];
          } else if (gx2) gx2[(int64_t)b * gbs2 + (int64_t)(c - s.C1) * P + p] = v;
        }
        float a = g * xh, d = g;
#pragma unroll
        for (int o = NPB / 2; o > 0; o >>= 1) {
          a += __shfl_xor(a, o, NPB);
          d += __shfl_xor(d, o, NPB);
        }
        if (px == 0) { pw[c] = a; pw[C + c] = d; }
      }
    }
  }
}

// ---- backward as two STREAMING kernels (round 3) ------------------------------------------------------------
// The fused kernels above move 128-byte row segments (32 pixels x 4 B per channel row) because a workgroup must see
// all C channels of its pixels: 3.4 TB/s forward, and the backward's second read of (gy, x) misses the caches in the
// training step (256 KiB per workgroup x 1024 resident workgroups).  Split by WHAT IS REDUCED instead:
//   stats : m1[b,p] = sum_c gy w / C,  m2[b,p] = sum_c gy w xhat / (C-1): lanes along the pixel axis (256-byte row
//           segments per wave), the four waves of a workgroup take channels g, g+4, ...; reads gy and x once;
//   apply : gx = rstd (gy w - m1 - xhat m2) (+ addend), one workgroup per (sample, channel) row, 16-byte accesses,
//           and - the row being one channel - the parameter-gradient sums of that row fall out of the same pass.
// Same HBM bytes as the reread kernel (gy and x twice), all of them at streaming efficiency; deterministic.
// GY16 (round 6, bf16-mixed mode): the cotangent is a bf16 tensor - the data gradient of the pointwise GEMM that consumed a
// bf16-stored y, bf16-valued in the reference's autocast backward as well (the gradient of conv2d's bf16 input).
constexpr int NSTAT_PX = 64, NSTAT_G = 4, NSTAT_U = 8;
template <bool GY16>
__global__ void __launch_bounds__(NSTAT_PX * NSTAT_G)
channel_norm_bwd_stats_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ w,
                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                              float* __restrict__ m1_out, float* __restrict__ m2_out, int P, int tiles) {
  __shared__ float red[2][NSTAT_G][NSTAT_PX];
  const int C = s.C1 + s.C2;
  const int b = blockIdx.x / tiles, p0 = (blockIdx.x - b * tiles) * NSTAT_PX;
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = min(p0 + lane, P - 1);          // clamped loads, predicated store
  const float mean = mean_in[(int64_t)b * P + p], rstd = rstd_in[(int64_t)b * P + p];
  const float* gyb = gy + (int64_t)b * C * P + p;
  [[maybe_unused]] const uint16_t* gyb16 = reinterpret_cast<const uint16_t*>(gy) + (int64_t)b * C * P + p;
  float s1 = 0.f, s2 = 0.f;
  for (int c0 = grp; c0 < C; c0 += NSTAT_G * NSTAT_U) {
    float gv[NSTAT_U], xv[NSTAT_U], wv[NSTAT_U];
#pragma unroll
    for (int j = 0; j < NSTAT_U; ++j) {
      const int c = min(c0 + NSTAT_G * j, C - 1);            // wave-uniform
      if constexpr (GY16) gv[j] = bf16_widen(gyb16[(int64_t)c * P]); else gv[j] = gyb[(int64_t)c * P];
      xv[j] = s.row(b, c, P)[p];
      wv[j] = c0 + NSTAT_G * j < C ? w[c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NSTAT_U; ++j) {
      const float gh = gv[j] * wv[j];
      s1 += gh;
      s2 += gh * ((xv[j] - mean) * rstd);
    }
  }
  red[0][grp][lane] = s1;
  red[1][grp][lane] = s2;
  __syncthreads();
  if (grp < 2 && p0 + lane < P) {
    const float t = (red[grp][0][lane] + red[grp][1][lane]) + (red[grp][2][lane] + red[grp][3][lane]);
    if (grp == 0) m1_out[(int64_t)b * P + p] = t / (float)C;
    else m2_out[(int64_t)b * P + p] = t / (float)(C - 1);
  }
}

// one workgroup per (b, c, chunk of `span` pixels); partial[(b * chunks + chunk)][2][C] = this chunk's sums of
// gy xhat and of gy for channel c (reduced by channel_norm_bwd_fused_finish / finish2 in a fixed order)
template <bool VEC, bool GY16 = false>
__global__ void __launch_bounds__(256)
channel_norm_bwd_apply_kernel(const float* __restrict__ gy, CatSrc s, const float* __restrict__ w,
                              const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                              const float* __restrict__ m1_in, const float* __restrict__ m2_in,
                              float* __restrict__ gx1, float* __restrict__ gx2, int64_t gbs1, int64_t gbs2,
                              const float* __restrict__ add1, int64_t abs1, float* __restrict__ partial,
                              int P, int span, int chunks) {
  __shared__ float red[2][4];
  const int C = s.C1 + s.C2;
  const int chunk = blockIdx.x % chunks, row = blockIdx.x / chunks;
  const int c = row % C, b = row / C;
  const int q0 = chunk * span, q1 = min(q0 + span, P);
  const float* g = gy + ((int64_t)b * C + c) * P;
  [[maybe_unused]] const uint16_t* g16 = reinterpret_cast<const uint16_t*>(gy) + ((int64_t)b * C + c) * P;
  const float* x = s.row(b, c, P);
  const bool first = c < s.C1;
  float* out = first ? gx1 + (int64_t)b * gbs1 + (int64_t)c * P
                     : (gx2 ? gx2 + (int64_t)b * gbs2 + (int64_t)(c - s.C1) * P : nullptr);
  const float* ad = (first && add1) ? add1 + (int64_t)b * abs1 + (int64_t)c * P : nullptr;
  const float* mean = mean_in + (int64_t)b * P;
  const float* rstd = rstd_in + (int64_t)b * P;
  const float* m1 = m1_in + (int64_t)b * P;
  const float* m2 = m2_in + (int64_t)b * P;
  const float wc = w[c];
  float a = 0.f, d = 0.f;
  if (VEC) {
    for (int q = q0 + 4 * (int)threadIdx.x; q < q1; q += 4 * 256) {
      float4 gv;
      if constexpr (GY16) {
        const uint2 r = *reinterpret_cast<const uint2*>(g16 + q);
        gv = make_float4(bf16_widen(r.x & 0xffffu), bf16_widen(r.x >> 16), bf16_widen(r.y & 0xffffu), bf16_widen(r.y >> 16));
      } else {
        gv = *reinterpret_cast<const float4*>(g + q);
      }
      const float4 xv = *reinterpret_cast<const float4*>(x + q);
      const float4 mu = *reinterpret_cast<const float4*>(mean + q), rs = *reinterpret_cast<const float4*>(rstd + q);
      const float4 a1 = *reinterpret_cast<const float4*>(m1 + q), a2 = *reinterpret_cast<const float4*>(m2 + q);
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ad) av = *reinterpret_cast<const float4*>(ad + q);
      const float xh0 = (xv.x - mu.x) * rs.x, xh1 = (xv.y - mu.y) * rs.y, xh2 = (xv.z - mu.z) * rs.z,
                  xh3 = (xv.w - mu.w) * rs.w;
      a += (gv.x * xh0 + gv.y * xh1) + (gv.z * xh2 + gv.w * xh3);
      d += (gv.x + gv.y) + (gv.z + gv.w);
      if (out) {
        float4 o;
        o.x = rs.x * (gv.x * wc - a1.x - xh0 * a2.x) + av.x;
        o.y = rs.y * (gv.y * wc - a1.y - xh1 * a2.y) + av.y;
        o.z = rs.z * (gv.z * wc - a1.z - xh2 * a2.z) + av.z;
        o.w = rs.w * (gv.w * wc - a1.w - xh3 * a2.w) + av.w;
        *reinterpret_cast<float4*>(out + q) = o;
      }
    }
  } else {
    for (int q = q0 + (int)threadIdx.x; q < q1; q += 256) {
      const float gv = GY16 ? bf16_widen(g16[q]) : g[q], xh = (x[q] - mean[q]) * rstd[q];
      a += gv * xh;
      d += gv;
      if (out) out[q] = rstd[q] * (gv * wc - m1[q] - xh * m2[q]) + (ad ? ad[q] : 0.f);
    }
  }
  a = wave_sum_dpp(a);
  d = wave_sum_dpp(d);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = d; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    partial[((int64_t)b * chunks + chunk) * 2 * C + (int64_t)threadIdx.x * C + c] = t;
  }
}

// Two-stage reduction of the per-block partial sums in a fixed order (no atomics: bit-reproducible):
// stage 1, grid (ceil(C/256), chunks): chunk[y][0][c] = sum over the chunk's rows of partial[k][0][c] (and [1]);
// stage 2, grid ceil(C/256): gw[c] = sum_y chunk[y][0][c], gb[c] = sum_y chunk[y][1][c].
__global__ void __launch_bounds__(256)
channel_norm_bwd_fused_finish(const float* __restrict__ partial, float* __restrict__ chunk, int C, int nblk,
                              int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int k0 = blockIdx.y * rows_per_chunk, k1 = min(k0 + rows_per_chunk, nblk);
  double a = 0.0, d = 0.0;   // the long sums run in double: what is left is the rounding of the 32-pixel block sums
  for (int k = k0; k < k1; ++k) {
    a += (double)partial[(int64_t)k * 2 * C + c];
    d += (double)partial[(int64_t)k * 2 * C + C + c];
  }
  chunk[(int64_t)blockIdx.y * 2 * C + c] = (float)a;
  chunk[(int64_t)blockIdx.y * 2 * C + C + c] = (float)d;
}

__global__ void __launch_bounds__(256)
channel_norm_bwd_fused_finish2(const float* __restrict__ chunk, float* __restrict__ gw, float* __restrict__ gb,
                               int C, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, d = 0.0;
  for (int y = 0; y < chunks; ++y) {
    a += (double)chunk[(int64_t)y * 2 * C + c];
    d += (double)chunk[(int64_t)y * 2 * C + C + c];
  }
  gw[c] = (float)a;
  gb[c] = (float)d;
}

int dw_chunks(int B, int C, int P) {
  const int64_t total = (int64_t)B * P;
  int chunks = (2048 + C - 1) / C;
  chunks = (int)std::min<int64_t>(chunks, (total + 255) / 256);
  return std::max(chunks, 1);
}

int g_norm_fwd_px = 32;   // pixels per forward workgroup (32 or 64); diagnostic knob
int g_norm_bwd_reread = 1;  // backward: re-read x through L2 (two workgroups per CU) vs xhat parked in LDS

int check_norm(const char* name, int B, int C1, int C2, int P) {
  PD_REQUIRE(B >= 0 && C1 >= 1 && C2 >= 0 && P >= 1, "%s: bad shape", name);
  PD_REQUIRE(C1 + C2 >= 2, "%s: needs at least two channels (unbiased variance)", name);
  PD_REQUIRE((int64_t)B * ((P + 31) / 32) < (1ll << 31), "%s: too large", name);
  return 0;
}

}  // namespace

#ifdef PARADIS_DEV_KNOBS   // development build only (`make dev`, tools/stencil_bench.py)
extern "C" void paradis_debug_set_norm_fwd_px(int px) { g_norm_fwd_px = px == 64 ? 64 : 32; }
extern "C" void paradis_debug_set_norm_bwd_reread(int on) { g_norm_bwd_reread = on ? 1 : 0; }
#endif

#ifndef NORM_FWD32      // (A/B builds: 0 = the generic 32-pixel kernel)
#define NORM_FWD32 1
#endif
template <bool Y16>
static int channel_norm_fwd_impl(const float* x1, const float* x2, const float* w,
                                 const float* b, float* y, float* mean, float* rstd, int B,
                                 int C1, int C2, int P, int64_t x1_bs, int64_t x2_bs, float eps, void* stream) {
  if (int e = check_norm("channel_norm_fwd", B, C1, C2, P)) return e;
  if (B == 0) return 0;       // (an empty batch has no x2 pointer either)
  PD_REQUIRE(C2 == 0 || x2 != nullptr, "channel_norm_fwd: x2 missing");
  CatSrc s{x1, x2, C1, C2, x1_bs, x2_bs};
  const int C = C1 + C2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(1024);
  if (C <= 32 * 36 && g_norm_fwd_px == 32) {          // 32 pixels x 32 groups, two workgroups per CU
    const int tiles = (P + 31) / 32;
    const dim3 grid((unsigned)((int64_t)B * tiles));
    if (C <= 32 * 4)
      hipLaunchKernelGGL((channel_norm_fwd_kernel<4, 32, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
    else if (NORM_FWD32 && (C1 & 1) == 0 && (C2 & 1) == 0 && (int64_t)2 * P < (1ll << 30))
      hipLaunchKernelGGL((channel_norm_fwd32_kernel<36, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
    else
      hipLaunchKernelGGL((channel_norm_fwd_kernel<36, 32, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
  } else {
    const int tiles = (P + NPX - 1) / NPX;
    const dim3 grid((unsigned)((int64_t)B * tiles));
    if (C <= 16 * 8)
      hipLaunchKernelGGL((channel_norm_fwd_kernel<8, NPX, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
    else if (C <= 16 * 72)
      hipLaunchKernelGGL((channel_norm_fwd_kernel<72, NPX, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
    else
      hipLaunchKernelGGL((channel_norm_fwd_kernel<0, NPX, Y16>), grid, block, 0, st, s, w, b, y, mean, rstd, P, tiles, eps);
  }
  PD_CHECK_LAUNCH("channel_norm_fwd");
  return 0;
}

extern "C" int paradis_channel_norm_fwd(const float* x1, const float* x2, const float* w,
                                        const float* b, float* y, float* mean, float* rstd, int B,
                                        int C1, int C2, int P, int64_t x1_bs, int64_t x2_bs, float eps, void* stream) {
  return channel_norm_fwd_impl<false>(x1, x2, w, b, y, mean, rstd, B, C1, C2, P, x1_bs, x2_bs, eps, stream);
}

// y written as bf16 [B][C1 + C2, P] (ABI 9; bf16-mixed mode: the consumer is a pointwise GEMM); mean / rstd stay fp32
extern "C" int paradis_channel_norm_fwd16(const float* x1, const float* x2, const float* w,
                                          const float* b, void* y, float* mean, float* rstd, int B,
                                          int C1, int C2, int P, int64_t x1_bs, int64_t x2_bs, float eps, void* stream) {
  return channel_norm_fwd_impl<true>(x1, x2, w, b, (float*)y, mean, rstd, B, C1, C2, P, x1_bs, x2_bs, eps, stream);
}

// pixels per workgroup of the apply kernel (one (sample, channel) row chunk); multiple of 4
constexpr int APPLY_SPAN = 8192;
#ifndef NORM_BWD_STREAMING    // (A/B builds: 0 = the fused reread kernel of round 2)
#define NORM_BWD_STREAMING 1
#endif

extern "C" size_t paradis_channel_norm_bwd_ws_bytes(int B, int C, int P) {
  const size_t b = (size_t)std::max(B, 1);
  const size_t three_kernel = (size_t)C * dw_chunks((int)b, C, P) * 2 * sizeof(float);
  const size_t nblk = b * ((P + 31) / 32);
  const size_t fused = (nblk + (nblk + 63) / 64) * 2 * C * sizeof(float);   // per-block partials + per-chunk sums
  const size_t nrow = b * ((P + APPLY_SPAN - 1) / APPLY_SPAN);
  const size_t streaming = (2 * b * P + (nrow + (nrow + 63) / 64) * 2 * C) * sizeof(float);
  return std::max({three_kernel, fused, streaming}) + 256;
}

// shapes whose backward runs as the two streaming kernels (the only ones with a bf16-cotangent instantiation)
static bool norm_bwd_streaming(int B, int C, int P) {
  const int apply_chunks = (P + APPLY_SPAN - 1) / APPLY_SPAN;
  return NORM_BWD_STREAMING && (int64_t)B * C * apply_chunks < (1ll << 31) && (int64_t)B * ((P + 63) / 64) < (1ll << 31);
}

template <bool GY16>
static int channel_norm_bwd_impl(const float* gy, const float* x1, const float* x2,
                                 const float* w, const float* mean, const float* rstd,
                                 float* gx1, float* gx2, float* gw, float* gb, int B, int C1,
                                 int C2, int P, int64_t x1_bs, int64_t x2_bs, int64_t gx1_bs,
                                 int64_t gx2_bs, const float* addend1, int64_t add1_bs,
                                 void* workspace, void* stream) {
  if (int e = check_norm("channel_norm_bwd", B, C1, C2, P)) return e;
  PD_REQUIRE(workspace != nullptr, "channel_norm_bwd: workspace required");
  const int C = C1 + C2;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (pd_zero_async(gw, C * sizeof(float), st) != hipSuccess ||
        pd_zero_async(gb, C * sizeof(float), st) != hipSuccess) {
      paradis_set_error("channel_norm_bwd: memset failed");
      return 2;
    }
    return 0;
  }
  CatSrc s{x1, x2, C1, C2, x1_bs, x2_bs};
  int nblk = 0;
  const int apply_chunks = (P + APPLY_SPAN - 1) / APPLY_SPAN;
  if (norm_bwd_streaming(B, C, P)) {
    float* m1 = (float*)workspace;
    float* m2 = m1 + (size_t)B * P;
    float* partial = m2 + (size_t)B * P;
    const int tiles = (P + NSTAT_PX - 1) / NSTAT_PX;
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec = P % 4 == 0 && x1_bs % 4 == 0 && (C2 == 0 || (x2_bs % 4 == 0 && a16(x2))) && gx1_bs % 4 == 0 &&
                     (gx2 == nullptr || (gx2_bs % 4 == 0 && a16(gx2))) && (addend1 == nullptr || (add1_bs % 4 == 0 && a16(addend1))) &&
                     a16(gy) && a16(x1) && a16(gx1) && a16(mean) && a16(rstd) && a16(workspace) && ((size_t)B * P) % 4 == 0;
    // (Round 6: the two passes over chunks of the batch, stats then apply, so that the apply pass would find the gy / x it
    //  re-reads in the 256 MB memory-side cache - every index is per sample, a chunk is the same launch on offset pointers.
    //  Measured, removed: 318 us per call in one chunk, 371 / 424 / 600 / 1338 us with read sets of 160 / 96 / 48 / 24 MB per
    //  chunk at 32 x 64, B = 32, C = 1024; the training step 152.7 -> 156.5 / 163.8 ms.  profiles/r06_norm_bwd_chunked.txt)
    hipLaunchKernelGGL(channel_norm_bwd_stats_kernel<GY16>, dim3((unsigned)((int64_t)B * tiles)), dim3(NSTAT_PX * NSTAT_G), 0,
                       st, gy, s, w, mean, rstd, m1, m2, P, tiles);
    const unsigned grid = (unsigned)((int64_t)B * C * apply_chunks);
    if (vec)
      hipLaunchKernelGGL((channel_norm_bwd_apply_kernel<true, GY16>), dim3(grid), dim3(256), 0, st, gy, s, w, mean, rstd,
                         (const float*)m1, (const float*)m2, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P,
                         APPLY_SPAN, apply_chunks);
    else
      hipLaunchKernelGGL((channel_norm_bwd_apply_kernel<false, GY16>), dim3(grid), dim3(256), 0, st, gy, s, w, mean, rstd,
                         (const float*)m1, (const float*)m2, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P,
                         APPLY_SPAN, apply_chunks);
    nblk = B * apply_chunks;
    const int rows = 64, chunks = (nblk + rows - 1) / rows;
    float* chunk = partial + (size_t)nblk * 2 * C;
    hipLaunchKernelGGL(channel_norm_bwd_fused_finish, dim3((C + 255) / 256, chunks), dim3(256), 0, st,
                       (const float*)partial, chunk, C, nblk, rows);
    hipLaunchKernelGGL(channel_norm_bwd_fused_finish2, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)chunk, gw,
                       gb, C, chunks);
    PD_CHECK_LAUNCH("channel_norm_bwd(streaming)");
    return 0;
  }
  if constexpr (GY16) {
    paradis_set_error("channel_norm_bwd16: shape outside the streaming kernels (paradis_channel_norm_bwd16_ok)");
    return 1;
  } else {
  if (C <= 32 * 36 && (int64_t)B * ((P + 31) / 32) < (1ll << 31)) {
    // g_norm_bwd_reread: 1 = stream-twice kernel (default; in the training step 231.6 vs 235.2 ms),
    // 0 = gy in registers + xhat in LDS, one workgroup per CU (half the HBM traffic, phases serialised).
    // (16-pixel tiles of the resident kernel, two workgroups per CU: 587 vs 365 us - not kept.)
    const int NPBr = 32;
    const int tiles32 = (P + NPBr - 1) / NPBr;
    nblk = B * tiles32;
    float* partial = (float*)workspace;
    const size_t lds = (size_t)(2 * 32 * NPBr + C * NPBr) * sizeof(float);
    static PerDeviceOnce once;
    if (once.first()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&channel_norm_bwd_fused_kernel<36, 32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(&channel_norm_bwd_fused_kernel<4, 32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        paradis_set_error("channel_norm_bwd: cannot reserve LDS");
        return 2;
      }
    }
    if (g_norm_bwd_reread == 1 && addend1)
      hipLaunchKernelGGL(channel_norm_bwd_reread_kernel<true>, dim3(nblk), dim3(512), 0, st, gy, s, w, mean,
                         rstd, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P, tiles32);
    else if (g_norm_bwd_reread == 1)
      hipLaunchKernelGGL(channel_norm_bwd_reread_kernel<false>, dim3(nblk), dim3(512), 0, st, gy, s, w, mean,
                         rstd, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P, tiles32);
    else if (C <= 32 * 4)
      hipLaunchKernelGGL((channel_norm_bwd_fused_kernel<4, 32>), dim3(nblk), dim3(32 * 32), lds, st, gy, s,
                         w, mean, rstd, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P, tiles32);
    else
      hipLaunchKernelGGL((channel_norm_bwd_fused_kernel<36, 32>), dim3(nblk), dim3(32 * 32), lds, st, gy, s,
                         w, mean, rstd, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, partial, P, tiles32);
    const int rows = 64, chunks = (nblk + rows - 1) / rows;
    float* chunk = partial + (size_t)nblk * 2 * C;
    hipLaunchKernelGGL(channel_norm_bwd_fused_finish, dim3((C + 255) / 256, chunks), dim3(256), 0, st, partial,
                       chunk, C, nblk, rows);
    hipLaunchKernelGGL(channel_norm_bwd_fused_finish2, dim3((C + 255) / 256), dim3(256), 0, st, chunk, gw, gb, C,
                       chunks);
    PD_CHECK_LAUNCH("channel_norm_bwd(fused)");
    return 0;
  }
  const int tiles = (P + NPX - 1) / NPX;
  hipLaunchKernelGGL(channel_norm_bwd_dx_kernel, dim3((unsigned)((int64_t)B * tiles)), dim3(1024), 0, st,
                     gy, s, w, mean, rstd, gx1, gx2, gx1_bs, gx2_bs, addend1, add1_bs, P, tiles);
  const int chunks = dw_chunks(B, C, P);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(channel_norm_bwd_dw_kernel, dim3(C * chunks), dim3(256), 0, st, gy, s, mean, rstd,
                     partial, B, P, chunks);
  hipLaunchKernelGGL(channel_norm_bwd_finish, dim3((C + 255) / 256), dim3(256), 0, st, partial, gw, gb, C,
                     chunks);
  PD_CHECK_LAUNCH("channel_norm_bwd");
  return 0;
  }
}

extern "C" int paradis_channel_norm_bwd(const float* gy, const float* x1, const float* x2,
                                        const float* w, const float* mean, const float* rstd,
                                        float* gx1, float* gx2, float* gw, float* gb, int B, int C1,
                                        int C2, int P, int64_t x1_bs, int64_t x2_bs, int64_t gx1_bs,
                                        int64_t gx2_bs, const float* addend1, int64_t add1_bs,
                                        void* workspace, void* stream) {
  return channel_norm_bwd_impl<false>(gy, x1, x2, w, mean, rstd, gx1, gx2, gw, gb, B, C1, C2, P, x1_bs, x2_bs, gx1_bs, gx2_bs,
                                      addend1, add1_bs, workspace, stream);
}

// gy as a bf16 tensor [B][C1 + C2, P] (ABI 9; bf16-mixed mode); everything else as paradis_channel_norm_bwd.  Only where
// paradis_channel_norm_bwd16_ok says so (the caller widens gy otherwise).
extern "C" int paradis_channel_norm_bwd16_ok(int B, int C, int P) { return norm_bwd_streaming(std::max(B, 1), C, P) ? 1 : 0; }
extern "C" int paradis_channel_norm_bwd16(const void* gy, const float* x1, const float* x2,
                                          const float* w, const float* mean, const float* rstd,
                                          float* gx1, float* gx2, float* gw, float* gb, int B, int C1,
                                          int C2, int P, int64_t x1_bs, int64_t x2_bs, int64_t gx1_bs,
                                          int64_t gx2_bs, const float* addend1, int64_t add1_bs,
                                          void* workspace, void* stream) {
  return channel_norm_bwd_impl<true>((const float*)gy, x1, x2, w, mean, rstd, gx1, gx2, gw, gb, B, C1, C2, P, x1_bs, x2_bs,
                                     gx1_bs, gx2_bs, addend1, add1_bs, workspace, stream);
}
