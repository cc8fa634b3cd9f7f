// a7 / a11 / a14: stencils on the virtual geocyclic halo (no padded tensor is materialised).
//
//   dwconv_geo   : depthwise k x k (reference model/blocks.py:101-113, model/paradis.py:189-190)
//   avgpool_geo  : 5x5 box mean with stride (reference model/blocks.py:57-71)
//   upsample_lonp: lon-periodic bilinear, align_corners=True (reference model/paradis.py:208-220)
//
// All HBM-bound: depthwise reads 4 B + writes 4 B per (channel, point) (+ halo re-reads served
// by L2); tiles of 32x64 outputs are staged once in LDS including the halo.
#include <algorithm>
#include "common.h"

namespace {

constexpr int TH = 32, TW = 64;  // output tile; 256 threads: lane -> column, wave -> 8-row strip
constexpr int RPT = 8;           // rows per thread

// Stage a (TH+K-1) x (TW+K-1) tile, flat over the 256 threads (a row-per-wave variant measured
// 25-30 % slower: the 68-wide rows leave most lanes of the second pass idle).  Loads are issued in
// batches from clamped, always-valid addresses and selected afterwards: a load under a per-lane
// condition makes the compiler wait for each one separately (one memory round trip per element).
template <int K, bool GEO>
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ src, int H, int W,
                                           int ty0, int tx0) {
  constexpr int P = (K - 1) / 2, LW = TW + K - 1, LH = TH + K - 1, N = LH * LW, BATCH = 5;
  for (int i0 = threadIdx.x; i0 < N; i0 += 256 * BATCH) {
    float val[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = min(i0 + 256 * j, N - 1);
      const int lr = i / LW, lc = i - lr * LW;
      const int ii = ty0 + lr - P, jj = tx0 + lc - P;
      bool valid;
      int r, c;
      if (GEO) {
        valid = ii < H + P && jj < W + P;
        geo_src(min(ii, H + P - 1), min(jj, W + P - 1), H, W, r, c);
      } else {
        valid = ii >= 0 && ii < H && jj >= 0 && jj < W;
        r = min(max(ii, 0), H - 1); c = min(max(jj, 0), W - 1);
      }
      const float v = src[(int64_t)r * W + c];
      val[j] = valid ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j)
      if (i0 + 256 * j < N) tile[i0 + 256 * j] = val[j];
  }
}

// the tile is the whole padded plane (W == TW, H <= TH, 16-byte aligned plane, even halo): plain
// 16-byte copy of the interior, only the halo ring through the index map (common.h)
template <int K>
__device__ __forceinline__ void stage_any(float* tile, const float* __restrict__ src, int H, int W, int ty0,
                                          int tx0, bool whole_vec4) {
  if (whole_vec4) stage_plane_vec4(tile, src, H, W, (K - 1) / 2);
  else stage_tile<K, true>(tile, src, H, W, ty0, tx0);
}

// The whole-plane case again (W == TW, H <= TH, k = 5: stage_plane_vec4's conditions), split into LOAD and STORE so
// that a workgroup walking several planes can have the next plane's loads in flight while it computes the current one:
// with one plane per workgroup a CU has loads outstanding only about half of the time (4.6 TB/s; Little's law with
// eight 8-KB planes per CU).  The per-thread cells - two float4 of the interior, two halo cells (destination in the
// tile, source in the plane) - do not depend on the plane and are computed once.
// a wave-uniform base address pinned to scalar registers plus an unsigned 32-bit BYTE offset per lane: the access is
// `global_load/store v, v_off, s[base:base+1]` - no 64-bit address arithmetic, no 64-bit addresses kept in registers
// (with typed indexing the compiler only finds this form for 4-byte elements)
typedef __attribute__((address_space(1))) char* ubase_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // (HIP's float4 struct cannot be read through an address-space pointer on the host pass)
__device__ __forceinline__ ubase_t uniform_base(const void* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (ubase_t)(((uint64_t)hi << 32) | lo);
}
// (the empty asm keeps the 32-bit offset opaque at the access: otherwise the loop optimiser widens it once to a 64-bit
//  per-thread address, carries that through the plane loop - two registers per access - and adds the base on the
//  vector unit)
template <typename T>
__device__ __forceinline__ T load_at(ubase_t b, unsigned off) {
  asm volatile("" : "+v"(off));
  return *(const __attribute__((address_space(1))) T*)(b + off);
}
template <typename T>
__device__ __forceinline__ void store_at(ubase_t b, unsigned off, T v) {
  asm volatile("" : "+v"(off));
  *(__attribute__((address_space(1))) T*)(b + off) = v;
}

template <int K>
struct PlaneStager {
  static constexpr int P = (K - 1) / 2, LW = TW + K - 1;
  unsigned vsrc[2], hsrc[2];
  int vdst[2], hdst[2];
  __device__ __forceinline__ void init(int H) {
    constexpr int W = TW, w4 = W / 4, hc = 2 * P;
    const int nvec = H * w4, nhalo_rows = 2 * P * LW, nhalo = nhalo_rows + H * hc, Hp = H + 2 * P;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = threadIdx.x + 256 * j, vc = min(v, nvec - 1);
      const int y = vc / w4, x4 = vc - y * w4;
      vsrc[j] = (unsigned)vc * 16u;                     // byte offsets
      vdst[j] = v < nvec ? (y + P) * LW + P + 4 * x4 : -1;
      const int kk = threadIdx.x + 256 * j, k = min(kk, nhalo - 1);
      int lr, lc;
      if (k < nhalo_rows) {
        const int rr = k / LW;
        lc = k - rr * LW;
        lr = rr < P ? rr : Hp - 2 * P + rr;
      } else {
        const int e = k - nhalo_rows, rr = e / hc, cc = e - rr * hc;
        lr = rr + P;
        lc = cc < P ? cc : W + cc;
      }
      int sr, sc;
      geo_src(lr - P, lc - P, H, W, sr, sc);
      hsrc[j] = (unsigned)(sr * W + sc) * 4u;
      hdst[j] = kk < nhalo ? lr * LW + lc : -1;
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ F, f32x4 (&q)[2], float (&hv)[2]) const {
    const ubase_t b = uniform_base(F);
#pragma unroll
    for (int j = 0; j < 2; ++j) { q[j] = load_at<f32x4>(b, vsrc[j]); hv[j] = load_at<float>(b, hsrc[j]); }
  }
  // the same plane stored as bf16 (round 6: the cotangent of a bf16-stored output): half the byte offsets.  load16 keeps the
  // RAW words (q.x, q.y = four bf16; hv = one, zero-extended by the load) so that nothing waits for the data at the fetch
  // site - the prefetch stays in flight under the previous plane's arithmetic - and store16 widens them on the way to LDS.
  __device__ __forceinline__ void load16(const uint16_t* __restrict__ F, f32x4 (&q)[2], float (&hv)[2]) const {
    const ubase_t b = uniform_base(F);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32x2 r = load_at<u32x2>(b, vsrc[j] >> 1);
      q[j].x = __uint_as_float(r.x); q[j].y = __uint_as_float(r.y);
      hv[j] = __uint_as_float((uint32_t)load_at<uint16_t>(b, hsrc[j] >> 1));
    }
  }
  __device__ __forceinline__ void store16(float* tile, const f32x4 (&q)[2], const float (&hv)[2]) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (vdst[j] >= 0) {
        const uint32_t lo = __float_as_uint(q[j].x), hi = __float_as_uint(q[j].y);
        float2* d = reinterpret_cast<float2*>(tile + vdst[j]);
        d[0] = make_float2(__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u));
        d[1] = make_float2(__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u));
      }
      if (hdst[j] >= 0) tile[hdst[j]] = __uint_as_float(__float_as_uint(hv[j]) << 16);
    }
  }
  __device__ __forceinline__ void store(float* tile, const f32x4 (&q)[2], const float (&hv)[2]) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (vdst[j] >= 0) {
        float2* d = reinterpret_cast<float2*>(tile + vdst[j]);      // 8-byte aligned (P, LW even)
        d[0] = make_float2(q[j].x, q[j].y);
        d[1] = make_float2(q[j].z, q[j].w);
      }
      if (hdst[j] >= 0) tile[hdst[j]] = hv[j];
    }
  }
};
#ifndef DWCONV_PLANE_CHUNK       // (A/B builds)
#define DWCONV_PLANE_CHUNK 4
#endif
constexpr int PLANE_CHUNK = DWCONV_PLANE_CHUNK;   // planes per workgroup on the whole-plane path
#ifndef DWCONV_PLANES            // (A/B builds: 0 = one plane per workgroup)
#define DWCONV_PLANES 1
#endif
#ifndef DWCONV_BWD_FUSED         // (A/B builds: 0 = paradis_dwconv_geo_bwd runs the two separate kernels)
#define DWCONV_BWD_FUSED 1
#endif

// FLIP=false: y = w (*) geo-padded x  (+bias).   FLIP=true: self-alias part of the data gradient.
template <int K, bool FLIP>
__device__ __forceinline__ void tile_stencil(const float* tile, const float* __restrict__ wc,
                                             float (&acc)[RPT]) {
  constexpr int LW = TW + K - 1;
  const int x = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * RPT;
  float w[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) w[i] = wc[FLIP ? (K * K - 1 - i) : i];
#pragma unroll
  for (int o = 0; o < RPT; ++o) acc[o] = 0.f;
#pragma unroll
  for (int rr = 0; rr < RPT + K - 1; ++rr) {
    float val[K];
#pragma unroll
    for (int b = 0; b < K; ++b) val[b] = tile[(r0 + rr) * LW + x + b];
#pragma unroll
    for (int a = 0; a < K; ++a) {
      const int o = rr - a;
      if (o >= 0 && o < RPT) {
#pragma unroll
        for (int b = 0; b < K; ++b) acc[o] += w[a * K + b] * val[b];
      }
    }
    // (row by row: left alone the scheduler hoists the LDS reads of many rows and the kernel sits at exactly 64
    //  registers with no room for the prefetched plane)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// whole-plane path: PLANE_CHUNK planes per workgroup, the next plane's loads in flight during the stencil
// Y16 (round 6, bf16-mixed mode): y written as bf16 (round to nearest even) for a consumer that is the pointwise GEMM of
// the same SepConv - the value that GEMM rounds its operand to (reference model/blocks.py:107-110 under autocast).
__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

template <int K, bool Y16 = false>
__global__ void __launch_bounds__(256, 5)   // (8 waves per SIMD = 64 registers spill the prefetched plane: 235 us instead of 99)
dwconv_geo_fwd_planes_kernel(const float* __restrict__ x, const float* __restrict__ w,
                             const float* __restrict__ bias, float* __restrict__ y, int C, int H, int64_t planes) {
  constexpr int W = TW;
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  const int xl = threadIdx.x & 63, r0l = (threadIdx.x >> 6) * RPT;
  const int64_t first = (int64_t)blockIdx.x * PLANE_CHUNK;
  const int n = (int)min((int64_t)PLANE_CHUNK, planes - first);
  PlaneStager<K> sg;
  sg.init(H);
  f32x4 q[2];
  float hv[2];
  sg.load(x + first * (int64_t)H * W, q, hv);
  for (int i = 0; i < n; ++i) {
    const int64_t plane = first + i;
    const int c = (int)(plane % C);
    sg.store(tile, q, hv);
    __syncthreads();
    if (i + 1 < n) sg.load(x + (plane + 1) * (int64_t)H * W, q, hv);
    float acc[RPT];
    tile_stencil<K, false>(tile, w + (int64_t)c * K * K, acc);
    const float bv = bias ? bias[c] : 0.f;
    constexpr int ES = Y16 ? 2 : 4;
    const ubase_t yp = uniform_base(reinterpret_cast<const char*>(y) + plane * (int64_t)H * W * ES);
    const unsigned o0 = (unsigned)(r0l * W + xl) * (unsigned)ES;
#pragma unroll
    for (int o = 0; o < RPT; ++o)
      if (r0l + o < H) {      // (the row step is on the scalar base)
        if constexpr (Y16) store_at<uint16_t>(yp + o * W * ES, o0, bf16_bits(acc[o] + bv));
        else store_at<float>(yp + o * W * ES, o0, acc[o] + bv);
      }
    __syncthreads();
  }
}

template <int K, bool Y16 = false>
__global__ void __launch_bounds__(256)
dwconv_geo_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                      const float* __restrict__ bias, float* __restrict__ y, int C, int H, int W,
                      int tiles_x, int tiles, int whole_vec4) {
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  const int64_t plane = blockIdx.x / tiles;
  const int t = blockIdx.x - plane * tiles;
  const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
  const int c = plane % C;
  stage_any<K>(tile, x + plane * (int64_t)H * W, H, W, ty0, tx0, whole_vec4);
  __syncthreads();
  float acc[RPT];
  tile_stencil<K, false>(tile, w + (int64_t)c * K * K, acc);
  const float bv = bias ? bias[c] : 0.f;
  const int xx = tx0 + (threadIdx.x & 63), r0 = ty0 + (threadIdx.x >> 6) * RPT;
  if (xx < W) {
    float* yp = y + plane * (int64_t)H * W;
    uint16_t* yp16 = reinterpret_cast<uint16_t*>(y) + plane * (int64_t)H * W;
#pragma unroll
    for (int o = 0; o < RPT; ++o)
      if (r0 + o < H) {
        const float v = acc[o] + bv;
        if constexpr (Y16) yp16[(int64_t)(r0 + o) * W + xx] = bf16_bits(v);
        else yp[(int64_t)(r0 + o) * W + xx] = v;
      }
  }
}

// Data gradient.  With the halo virtual, gx = PadAdjoint(ConvTranspose(gy)).  Folding the halo
// aliases back analytically gives a stencil on the *geocyclic extension* E of gy itself:
//   rows of E inside the image : transposed taps        w[p-dr][p-dc]
//   rows of E beyond a pole    : row index NOT flipped   w[p+dr][p-dc]  (the over-the-pole glide
//                                reflection reverses the row direction), and they only feed source
//                                rows 1..p (south) / H-1-p..H-2 (north);
//   the pole row itself seen through the mirror (dr = -y resp. H-1-y) needs the W/2-shifted pole
//   row, which differs from E's unshifted row: K extra taps read from global memory.
// Longitude wrap is implied by E's periodic columns.
// `addend` (nullable): gx = dgrad + addend - the other gradient of the stencil's input (a consumer around the block:
// the gated blend's share of the advection input), added here instead of by a separate pass of the autograd engine.
template <int K>
__global__ void __launch_bounds__(256)
dwconv_geo_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w, const float* __restrict__ addend,
                        float* __restrict__ gx, int C, int H, int W, int tiles_x, int tiles,
                        int whole_vec4) {
  constexpr int P = (K - 1) / 2, LW = TW + K - 1;
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  const int64_t plane = blockIdx.x / tiles;
  const int t = blockIdx.x - plane * tiles;
  const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
  const int c = plane % C;
  const float* g = gy + plane * (int64_t)H * W;
  const float* wc = w + (int64_t)c * K * K;
  stage_any<K>(tile, g, H, W, ty0, tx0, whole_vec4);
  __syncthreads();
  const int xl = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * RPT;
  float wr[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wr[i] = wc[i];
  float acc[RPT];
#pragma unroll
  for (int o = 0; o < RPT; ++o) acc[o] = 0.f;
#pragma unroll
  for (int rr = 0; rr < RPT + K - 1; ++rr) {
    const int ii = ty0 + r0 + rr - P;   // image row of this tile row (wave-uniform)
    float val[K];
#pragma unroll
    for (int b = 0; b < K; ++b) val[b] = tile[(r0 + rr) * LW + xl + b];
    if (ii >= 0 && ii < H) {
#pragma unroll
      for (int a = 0; a < K; ++a) {       // a = tile row offset of output o: rr = o + a, dr = a - P
        const int o = rr - a;
        if (o >= 0 && o < RPT) {
#pragma unroll
          for (int b = 0; b < K; ++b) acc[o] += wr[(K - 1 - a) * K + (K - 1 - b)] * val[b];
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < K; ++a) {
        const int o = rr - a;
        if (o >= 0 && o < RPT) {
          const int yy = ty0 + r0 + o;
          const bool feeds = (ii < 0) ? (yy >= 1) : (yy <= H - 2);
          if (feeds) {
#pragma unroll
            for (int b = 0; b < K; ++b) acc[o] += wr[a * K + (K - 1 - b)] * val[b];
          }
        }
      }
    }
  }
  const int xx = tx0 + xl;
  if (xx >= W) return;
  const int half = W >> 1;
#pragma unroll
  for (int o = 0; o < RPT; ++o) {
    const int yy = ty0 + r0 + o;
    if (yy >= H) break;
    float extra = 0.f;
    // mirrored pole rows: E'[0][jj] = gy[0][jj + W/2], E'[H-1][jj] = gy[H-1][jj + W/2].  When the
    // plane is a single tile both pole rows are in LDS (tile rows P and H-1+P, column + P).
    const bool south = yy >= 1 && yy <= P, north = yy >= H - 1 - P && yy <= H - 2;
    if (south || north) {
      const int a = south ? P - yy : P + (H - 1 - yy);   // dr = -yy  resp.  H-1-yy
      const int prow = south ? 0 : H - 1;
#pragma unroll
      for (int b = 0; b < K; ++b) {     // dc = P - b
        int col = xx + P - b + half;
        if (col >= W) col -= W;          // xx + P - b + W/2 lies in [-P+W/2, W + P + W/2)
        if (col >= W) col -= W;
        const float pv = tiles == 1 ? tile[(prow + P) * LW + col + P] : g[(int64_t)prow * W + col];
        extra += wc[a * K + b] * pv;   // (wc, not the register copy: a is not a compile-time index)
      }
    }
    const int64_t at = plane * (int64_t)H * W + (int64_t)yy * W + xx;
    gx[at] = acc[o] + extra + (addend ? addend[at] : 0.f);
  }
}

// whole-plane path of the data gradient: PLANE_CHUNK planes per workgroup, next plane's loads in flight (see
// dwconv_geo_fwd_planes_kernel); the mirrored pole rows come from the tile
template <int K, bool ADD>
__global__ void __launch_bounds__(256, 5)
dwconv_geo_dgrad_planes_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                               const float* __restrict__ addend, float* __restrict__ gx,
                               int C, int H, int64_t planes) {
  constexpr int P = (K - 1) / 2, LW = TW + K - 1, W = TW;
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  const int xl = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * RPT;
  const int64_t first = (int64_t)blockIdx.x * PLANE_CHUNK;
  const int n = (int)min((int64_t)PLANE_CHUNK, planes - first);
  PlaneStager<K> sg;
  sg.init(H);
  f32x4 q[2];
  float hv[2];
  sg.load(gy + first * (int64_t)H * W, q, hv);
  for (int i = 0; i < n; ++i) {
    const int64_t plane = first + i;
    const float* wc = w + (int64_t)(plane % C) * K * K;
    sg.store(tile, q, hv);
    __syncthreads();
    if (i + 1 < n) sg.load(gy + (plane + 1) * (int64_t)H * W, q, hv);
    const unsigned o0 = (unsigned)(r0 * W + xl) * 4u;
    float av[RPT];                  // this plane's addend values: in flight under the stencil arithmetic
    if (ADD) {
      const ubase_t ab = uniform_base(addend + plane * (int64_t)H * W);
#pragma unroll
      for (int o = 0; o < RPT; ++o) av[o] = (r0 + o < H) ? load_at<float>(ab + o * W * 4, o0) : 0.f;
    }
    float wr[K * K];
#pragma unroll
    for (int j = 0; j < K * K; ++j) wr[j] = wc[j];
    float acc[RPT];
#pragma unroll
    for (int o = 0; o < RPT; ++o) acc[o] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPT + K - 1; ++rr) {
      const int ii = r0 + rr - P;   // image row of this tile row (wave-uniform)
      float val[K];
#pragma unroll
      for (int b = 0; b < K; ++b) val[b] = tile[(r0 + rr) * LW + xl + b];
      if (ii >= 0 && ii < H) {
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const int o = rr - a;
          if (o >= 0 && o < RPT) {
#pragma unroll
            for (int b = 0; b < K; ++b) acc[o] += wr[(K - 1 - a) * K + (K - 1 - b)] * val[b];
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const int o = rr - a;
          if (o >= 0 && o < RPT) {
            const int yy = r0 + o;
            const bool feeds = (ii < 0) ? (yy >= 1) : (yy <= H - 2);
            if (feeds) {
#pragma unroll
              for (int b = 0; b < K; ++b) acc[o] += wr[a * K + (K - 1 - b)] * val[b];
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const ubase_t gp = uniform_base(gx + plane * (int64_t)H * W);
    constexpr int half = W >> 1;
#pragma unroll
    for (int o = 0; o < RPT; ++o) {
      const int yy = r0 + o;
      if (yy < H) {
        float extra = 0.f;
        const bool south = yy >= 1 && yy <= P, north = yy >= H - 1 - P && yy <= H - 2;
        if (south || north) {
          const int a = south ? P - yy : P + (H - 1 - yy);   // dr = -yy  resp.  H-1-yy
          const int prow = south ? 0 : H - 1;
#pragma unroll
          for (int b = 0; b < K; ++b) {     // dc = P - b
            int col = xl + P - b + half;
            if (col >= W) col -= W;
            if (col >= W) col -= W;
            extra += wc[a * K + b] * tile[(prow + P) * LW + col + P];
          }
        }
        store_at<float>(gp + o * W * 4, o0, ADD ? (acc[o] + extra) + av[o] : acc[o] + extra);   // (= the two-pass sum, bit for bit)
      }
    }
    __syncthreads();
  }
}

// partial[c][chunk][K*K (+1 for bias)] ; items of a channel = (batch n, tile t)
template <int K>
__global__ void __launch_bounds__(256)
dwconv_geo_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                        float* __restrict__ partial, int B, int C, int H, int W, int tiles_x,
                        int tiles, int chunks, int whole_vec4) {
  constexpr int LW = TW + K - 1, NW = K * K + 1;
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  __shared__ float red[4][NW];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x - c * chunks;
  const int items = B * tiles;
  const int xl = threadIdx.x & 63, wave = threadIdx.x >> 6, r0l = wave * RPT;
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
  float gsum = 0.f;
  // (tile cells no staging path writes meet zero cotangents: keep 0 x NaN out, see the planes kernel below)
  for (int i = threadIdx.x; i < (TH + K - 1) * (TW + K - 1); i += 256) tile[i] = 0.f;
  for (int item = chunk; item < items; item += chunks) {
    const int n = item / tiles, t = item - n * tiles;
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const int64_t plane = (int64_t)n * C + c;
    __syncthreads();
    stage_any<K>(tile, x + plane * (int64_t)H * W, H, W, ty0, tx0, whole_vec4);
    __syncthreads();
    float g[RPT];
    const int xx = tx0 + xl;
#pragma unroll
    for (int o = 0; o < RPT; ++o) {
      const int yy = ty0 + r0l + o;
      g[o] = (xx < W && yy < H) ? gy[plane * (int64_t)H * W + (int64_t)yy * W + xx] : 0.f;
      gsum += g[o];
    }
#pragma unroll
    for (int rr = 0; rr < RPT + K - 1; ++rr) {
      float val[K];
#pragma unroll
      for (int b = 0; b < K; ++b) val[b] = tile[(r0l + rr) * LW + xl + b];
#pragma unroll
      for (int a = 0; a < K; ++a) {
        const int o = rr - a;
        if (o >= 0 && o < RPT) {
#pragma unroll
          for (int b = 0; b < K; ++b) acc[a * K + b] += g[o] * val[b];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    float s = wave_sum_dpp(acc[i]);
    if (xl == 0) red[wave][i] = s;
  }
  {
    float s = wave_sum_dpp(gsum);
    if (xl == 0) red[wave][K * K] = s;
  }
  __syncthreads();
  if (threadIdx.x < NW) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[((int64_t)c * chunks + chunk) * NW + threadIdx.x] = s;
  }
}

// whole-plane path of the weight gradient: the items of a chunk are whole planes (sample n, channel c); the next
// item's x plane and gy rows are loaded while the current one is accumulated
template <int K>
__global__ void __launch_bounds__(256, 5)
dwconv_geo_wgrad_planes_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                               float* __restrict__ partial, int B, int C, int H, int chunks) {
  constexpr int LW = TW + K - 1, NW = K * K + 1, W = TW;
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  __shared__ float red[4][NW];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x - c * chunks;
  const int xl = threadIdx.x & 63, wave = threadIdx.x >> 6, r0l = wave * RPT;
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
  float gsum = 0.f;
  PlaneStager<K> sg;
  sg.init(H);
  f32x4 q[2];
  float hv[2], gn[RPT];
  const unsigned g0 = (unsigned)(r0l * W + xl) * 4u;
  auto fetch = [&](int item) __attribute__((always_inline)) {
    const int64_t off = ((int64_t)item * C + c) * (int64_t)H * W;
    sg.load(x + off, q, hv);
    const ubase_t gb = uniform_base(gy + off);
#pragma unroll
    for (int o = 0; o < RPT; ++o) gn[o] = (r0l + o < H) ? load_at<float>(gb + o * W * 4, g0) : 0.f;
  };
  // rows of the tile beyond the padded plane (H < 32: a wave's strip may overshoot) are never staged: they meet
  // cotangent rows that are zero, and 0 x uninitialised LDS could be 0 x NaN - define them once
  for (int i = threadIdx.x; i < (TH + K - 1) * (TW + K - 1); i += 256) tile[i] = 0.f;
  __syncthreads();
  if (chunk < B) fetch(chunk);
  for (int item = chunk; item < B; item += chunks) {
    sg.store(tile, q, hv);
    float g[RPT];
#pragma unroll
    for (int o = 0; o < RPT; ++o) { g[o] = gn[o]; gsum += g[o]; }
    __syncthreads();
    if (item + chunks < B) fetch(item + chunks);
#pragma unroll
    for (int rr = 0; rr < RPT + K - 1; ++rr) {
      float val[K];
#pragma unroll
      for (int b = 0; b < K; ++b) val[b] = tile[(r0l + rr) * LW + xl + b];
#pragma unroll
      for (int a = 0; a < K; ++a) {
        const int o = rr - a;
        if (o >= 0 && o < RPT) {
#pragma unroll
          for (int b = 0; b < K; ++b) acc[a * K + b] += g[o] * val[b];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    float s = wave_sum_dpp(acc[i]);
    if (xl == 0) red[wave][i] = s;
  }
  {
    float s = wave_sum_dpp(gsum);
    if (xl == 0) red[wave][K * K] = s;
  }
  __syncthreads();
  if (threadIdx.x < NW) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[((int64_t)c * chunks + chunk) * NW + threadIdx.x] = s;
  }
}

// Data gradient AND weight gradient of the whole-plane path in one pass (round 4): both read the cotangent plane; run
// apart they move gy twice (812 MB per 32x64 B=32 layer at C = 1024), together once (603 MB).  The workgroup is the
// weight-gradient kernel's - one channel, the samples of a chunk, the 26 sums in registers across planes - and stages
// TWO tiles per plane: gy with its geocyclic extension (what the data gradient convolves) and x with its halo; the
// cotangent values the weight gradient multiplies are the centre of the gy tile.  Same arithmetic in the same order as
// dwconv_geo_dgrad_planes_kernel and dwconv_geo_wgrad_planes_kernel: bit-identical results.
// GY16 (round 6, bf16-mixed mode): gy is a bf16 tensor - the data gradient of the SepConv's pointwise GEMM, which consumed
// the stencil's bf16-stored output (bf16-valued in the reference's autocast backward too).
template <int K, bool ADD, bool GY16 = false>
__global__ void __launch_bounds__(256, 4)
dwconv_geo_bwd_planes_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w,
                             const float* __restrict__ addend, float* __restrict__ gx, float* __restrict__ partial,
                             int B, int C, int H, int chunks) {
  constexpr int P = (K - 1) / 2, LW = TW + K - 1, NW = K * K + 1, W = TW, TN = (TH + K - 1) * (TW + K - 1);
  __shared__ float tg[TN], tx[TN];
  __shared__ float red[4][NW];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x - c * chunks;
  const int xl = threadIdx.x & 63, wave = threadIdx.x >> 6, r0 = wave * RPT;
  const float* wc = w + (int64_t)c * K * K;
  float accw[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) accw[i] = 0.f;
  float gsum = 0.f;
  PlaneStager<K> sg;
  sg.init(H);
  f32x4 qg[2], qx[2];
  float hg[2], hx[2];
  const unsigned o0 = (unsigned)(r0 * W + xl) * 4u;
  auto fetch = [&](int item) __attribute__((always_inline)) {
    const int64_t off = ((int64_t)item * C + c) * (int64_t)H * W;
    if constexpr (GY16) sg.load16(reinterpret_cast<const uint16_t*>(gy) + off, qg, hg);
    else sg.load(gy + off, qg, hg);
    sg.load(x + off, qx, hx);
  };
  // (rows of the tiles beyond the padded plane are never staged: define them once - see the weight-gradient kernel)
  for (int i = threadIdx.x; i < TN; i += 256) { tg[i] = 0.f; tx[i] = 0.f; }
  __syncthreads();
  if (chunk < B) fetch(chunk);
  for (int item = chunk; item < B; item += chunks) {
    const int64_t off = ((int64_t)item * C + c) * (int64_t)H * W;
    if constexpr (GY16) sg.store16(tg, qg, hg); else sg.store(tg, qg, hg);
    sg.store(tx, qx, hx);
    __syncthreads();
    if (item + chunks < B) fetch(item + chunks);
    float av[RPT];
    if (ADD) {
      const ubase_t ab = uniform_base(addend + off);
#pragma unroll
      for (int o = 0; o < RPT; ++o) av[o] = (r0 + o < H) ? load_at<float>(ab + o * W * 4, o0) : 0.f;
    }
    // ---- data gradient of this plane (dwconv_geo_dgrad_planes_kernel)
    {
      float wr[K * K];
#pragma unroll
      for (int j = 0; j < K * K; ++j) wr[j] = wc[j];
      float acc[RPT];
#pragma unroll
      for (int o = 0; o < RPT; ++o) acc[o] = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPT + K - 1; ++rr) {
        const int ii = r0 + rr - P;   // image row of this tile row (wave-uniform)
        float val[K];
#pragma unroll
        for (int b = 0; b < K; ++b) val[b] = tg[(r0 + rr) * LW + xl + b];
        if (ii >= 0 && ii < H) {
#pragma unroll
          for (int a = 0; a < K; ++a) {
            const int o = rr - a;
            if (o >= 0 && o < RPT) {
#pragma unroll
              for (int b = 0; b < K; ++b) acc[o] += wr[(K - 1 - a) * K + (K - 1 - b)] * val[b];
            }
          }
        } else {
#pragma unroll
          for (int a = 0; a < K; ++a) {
            const int o = rr - a;
            if (o >= 0 && o < RPT) {
              const int yy = r0 + o;
              const bool feeds = (ii < 0) ? (yy >= 1) : (yy <= H - 2);
              if (feeds) {
#pragma unroll
                for (int b = 0; b < K; ++b) acc[o] += wr[a * K + (K - 1 - b)] * val[b];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const ubase_t gp = uniform_base(gx + off);
      constexpr int half = W >> 1;
#pragma unroll
      for (int o = 0; o < RPT; ++o) {
        const int yy = r0 + o;
        if (yy < H) {
          float extra = 0.f;
          const bool south = yy >= 1 && yy <= P, north = yy >= H - 1 - P && yy <= H - 2;
          if (south || north) {
            const int a = south ? P - yy : P + (H - 1 - yy);
            const int prow = south ? 0 : H - 1;
#pragma unroll
            for (int b = 0; b < K; ++b) {
              int col = xl + P - b + half;
              if (col >= W) col -= W;
              if (col >= W) col -= W;
              extra += wc[a * K + b] * tg[(prow + P) * LW + col + P];
            }
          }
          store_at<float>(gp + o * W * 4, o0, ADD ? (acc[o] + extra) + av[o] : acc[o] + extra);
        }
      }
    }
    // ---- weight gradient: this plane's share of the channel's sums (dwconv_geo_wgrad_planes_kernel)
    {
      float g[RPT];
#pragma unroll
      for (int o = 0; o < RPT; ++o) {
        g[o] = (r0 + o < H) ? tg[(r0 + o + P) * LW + xl + P] : 0.f;
        gsum += g[o];
      }
#pragma unroll
      for (int rr = 0; rr < RPT + K - 1; ++rr) {
        float val[K];
#pragma unroll
        for (int b = 0; b < K; ++b) val[b] = tx[(r0 + rr) * LW + xl + b];
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const int o = rr - a;
          if (o >= 0 && o < RPT) {
#pragma unroll
            for (int b = 0; b < K; ++b) accw[a * K + b] += g[o] * val[b];
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    float s = wave_sum_dpp(accw[i]);
    if (xl == 0) red[wave][i] = s;
  }
  {
    float s = wave_sum_dpp(gsum);
    if (xl == 0) red[wave][K * K] = s;
  }
  __syncthreads();
  if (threadIdx.x < NW) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[((int64_t)c * chunks + chunk) * NW + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------- staged tiles of the larger grids
// Grids of more than one tile (128x256, 721x1440, ...; k = 5): the whole-plane kernels' structure - 16-byte loads from
// per-thread offsets computed once, the next item's loads in flight under the current item's arithmetic - on 32x64
// tiles that are always FULL: the last tile row / column starts at H - 32 / W - 64, overlapping its neighbour, and
// owns (stores, and counts in the weight gradient) only the rows / columns its neighbour does not.  A full tile has
// one fixed shape: 512 aligned float4 of interior and a 400-cell halo ring through the index map, two of each per
// thread, whatever the position.
template <int K>
struct TileStager {
  static constexpr int P = (K - 1) / 2, LW = TW + K - 1;
  unsigned vsrc[2], hsrc[2];
  int vdst[2], hdst[2];
  __device__ __forceinline__ void init(int H, int W, int ty0, int tx0) {
    constexpr int w4 = TW / 4, hc = 2 * P, nhalo_rows = 2 * P * LW, nhalo = nhalo_rows + TH * hc;
    static_assert(TH * w4 == 512 && nhalo <= 512, "two interior vectors and two halo cells per thread");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = threadIdx.x + 256 * j;
      const int y = v / w4, x4 = v - y * w4;
      vsrc[j] = (unsigned)((ty0 + y) * W + tx0 + 4 * x4) * 4u;          // byte offsets in the plane
      vdst[j] = (y + P) * LW + P + 4 * x4;
      const int kk = threadIdx.x + 256 * j, k = min(kk, nhalo - 1);
      int lr, lc;
      if (k < nhalo_rows) {
        const int rr = k / LW;
        lc = k - rr * LW;
        lr = rr < P ? rr : TH + rr;
      } else {
        const int e = k - nhalo_rows, rr = e / hc, cc = e - rr * hc;
        lr = rr + P;
        lc = cc < P ? cc : TW + cc;
      }
      int sr, sc;
      geo_src(ty0 + lr - P, tx0 + lc - P, H, W, sr, sc);
      hsrc[j] = (unsigned)(sr * W + sc) * 4u;
      hdst[j] = kk < nhalo ? lr * LW + lc : -1;
    }
  }
  __device__ __forceinline__ void load(const float* __restrict__ F, f32x4 (&q)[2], float (&hv)[2]) const {
    const ubase_t b = uniform_base(F);
#pragma unroll
    for (int j = 0; j < 2; ++j) { q[j] = load_at<f32x4>(b, vsrc[j]); hv[j] = load_at<float>(b, hsrc[j]); }
  }
  __device__ __forceinline__ void store(float* tile, const f32x4 (&q)[2], const float (&hv)[2]) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float2* d = reinterpret_cast<float2*>(tile + vdst[j]);      // 8-byte aligned (P, LW even)
      d[0] = make_float2(q[j].x, q[j].y);
      d[1] = make_float2(q[j].z, q[j].w);
      if (hdst[j] >= 0) tile[hdst[j]] = hv[j];
    }
  }
};

// position of tile t: (ty0, tx0) = where it is staged from, (ny0, nx0) = the first row / column it owns
struct TilePos { int ty0, tx0, ny0, nx0; };
__device__ __forceinline__ TilePos tile_pos(int t, int tiles_x, int H, int W) {
  const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
  TilePos p;
  p.ny0 = tyi * TH; p.nx0 = txi * TW;
  p.ty0 = min(p.ny0, H - TH); p.tx0 = min(p.nx0, W - TW);
  return p;
}

// consecutive workgroup ids go round the eight XCDs; hand every XCD a contiguous range of logical ids instead: the
// tiles of a plane chunk - neighbours that share halo cells - then run behind one L2
__device__ __forceinline__ int xcd_contiguous(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// forward: a workgroup walks PLANE_CHUNK planes at one tile position
template <int K, bool Y16 = false>
__global__ void __launch_bounds__(256, 5)
dwconv_geo_fwd_tiles_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                            float* __restrict__ y, int C, int H, int W, int tiles_x, int tiles, int64_t planes) {
  __shared__ float tile[(TH + K - 1) * (TW + K - 1)];
  const int L = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
  const int chunk = L / tiles, t = L - chunk * tiles;
  const TilePos tp = tile_pos(t, tiles_x, H, W);
  const int xl = threadIdx.x & 63, r0l = (threadIdx.x >> 6) * RPT;
  const int64_t first = (int64_t)chunk * PLANE_CHUNK, PS = (int64_t)H * W;
  const int n = (int)min((int64_t)PLANE_CHUNK, planes - first);
  TileStager<K> sg;
  sg.init(H, W, tp.ty0, tp.tx0);
  f32x4 q[2];
  float hv[2];
  sg.load(x + first * PS, q, hv);
  constexpr int ES = Y16 ? 2 : 4;
  const unsigned o0 = (unsigned)((tp.ty0 + r0l) * W + tp.tx0 + xl) * (unsigned)ES;
  const bool col_owned = tp.tx0 + xl >= tp.nx0;
  const int own0 = tp.ny0 - tp.ty0 - r0l;           // rows o >= own0 of this thread's strip are owned
  for (int i = 0; i < n; ++i) {
    const int64_t plane = first + i;
    const int c = (int)(plane % C);
    sg.store(tile, q, hv);
    __syncthreads();
    if (i + 1 < n) sg.load(x + (plane + 1) * PS, q, hv);
    float acc[RPT];
    tile_stencil<K, false>(tile, w + (int64_t)c * K * K, acc);
    const float bv = bias ? bias[c] : 0.f;
    const ubase_t yp = uniform_base(reinterpret_cast<const char*>(y) + plane * PS * ES);
#pragma unroll
    for (int o = 0; o < RPT; ++o)
      if (col_owned && o >= own0) {
        if constexpr (Y16) store_at<uint16_t>(yp + (int64_t)o * W * ES, o0, bf16_bits(acc[o] + bv));
        else store_at<float>(yp + (int64_t)o * W * ES, o0, acc[o] + bv);
      }
    __syncthreads();
  }
}

// item -> (tile, sample) of a channel's B x tiles items.  Tile fastest: a workgroup's contiguous range of items walks
// the tiles of one plane in row-major order, so the halo cells a tile shares with its left neighbour were read by the
// same CU one item earlier (L2 hits); sample fastest would keep the stager's offsets across items instead.
#ifndef DWCONV_BWD_TFAST         // (A/B builds)
#define DWCONV_BWD_TFAST 1
#endif
__device__ __forceinline__ void item_of(int item, int B, int tiles, int& t, int& n) {
  if (DWCONV_BWD_TFAST) { n = item / tiles; t = item - n * tiles; }
  else { t = item / B; n = item - t * B; }
}

// both gradients in one pass (dwconv_geo_bwd_planes_kernel's arithmetic on tiles): workgroup (channel c, chunk) walks a
// contiguous range of the channel's B x tiles items (item_of); the 26 sums stay in registers across items.  The
// standalone entry points run the same kernel with one half compiled out: whichever way the gradients are asked for,
// the bits are the same.  (The data gradient is also the one-tile-per-workgroup kernel's sum in the same order; the
// weight gradient partitions its sum differently from dwconv_geo_wgrad_kernel: same terms, another fixed order.)
template <int K, bool ADD, bool DG, bool WG>   // DG: data gradient, WG: weight gradient (either alone = the standalone entry points)
__global__ void __launch_bounds__(256, 4)
dwconv_geo_bwd_tiles_kernel(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ w,
                            const float* __restrict__ addend, float* __restrict__ gx, float* __restrict__ partial,
                            int B, int C, int H, int W, int tiles_x, int tiles, int chunks, int per) {
  constexpr int P = (K - 1) / 2, LW = TW + K - 1, NW = K * K + 1, TN = (TH + K - 1) * (TW + K - 1);
  __shared__ float tg[TN], tx[TN];
  __shared__ float red[4][NW];
  const int c = blockIdx.x / chunks, chunk = blockIdx.x - c * chunks;
  const int items = B * tiles, i0 = chunk * per, i1 = min(items, i0 + per);
  const int xl = threadIdx.x & 63, wave = threadIdx.x >> 6, r0 = wave * RPT;
  const float* wc = w + (int64_t)c * K * K;
  const int64_t PS = (int64_t)H * W;
  const int half = W >> 1;
  float accw[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) accw[i] = 0.f;
  float gsum = 0.f;
  TileStager<K> sg;
  f32x4 qg[2], qx[2];
  float hg[2], hx[2];
  int staged_t = -1;
  auto fetch = [&](int item) __attribute__((always_inline)) {
    int t, n;
    item_of(item, B, tiles, t, n);
    if (t != staged_t) {                                       // (wave-uniform)
      const TilePos np = tile_pos(t, tiles_x, H, W);
      sg.init(H, W, np.ty0, np.tx0);
      staged_t = t;
    }
    const int64_t off = ((int64_t)n * C + c) * PS;
    sg.load(gy + off, qg, hg);
    if (WG) sg.load(x + off, qx, hx);
  };
  if (i0 < i1) fetch(i0);
  for (int item = i0; item < i1; ++item) {
    int t, n;
    item_of(item, B, tiles, t, n);
    const TilePos tp = tile_pos(t, tiles_x, H, W);
    const int64_t off = ((int64_t)n * C + c) * PS;
    const float* gpl = gy + off;
    sg.store(tg, qg, hg);
    if (WG) sg.store(tx, qx, hx);
    __syncthreads();
    if (item + 1 < i1) fetch(item + 1);
    const unsigned o0 = (unsigned)((tp.ty0 + r0) * W + tp.tx0 + xl) * 4u;
    const int xx = tp.tx0 + xl;
    const bool col_owned = xx >= tp.nx0;
    const int own0 = tp.ny0 - tp.ty0 - r0;
    float av[RPT];
    if (ADD) {
      const ubase_t ab = uniform_base(addend + off);
#pragma unroll
      for (int o = 0; o < RPT; ++o) av[o] = load_at<float>(ab + (int64_t)o * W * 4, o0);
    }
    // ---- data gradient of this tile (dwconv_geo_dgrad_kernel)
    if constexpr (DG) {
      float wr[K * K];
#pragma unroll
      for (int j = 0; j < K * K; ++j) wr[j] = wc[j];
      float acc[RPT];
#pragma unroll
      for (int o = 0; o < RPT; ++o) acc[o] = 0.f;
#pragma unroll
      for (int rr = 0; rr < RPT + K - 1; ++rr) {
        const int ii = tp.ty0 + r0 + rr - P;   // image row of this tile row (wave-uniform)
        float val[K];
#pragma unroll
        for (int b = 0; b < K; ++b) val[b] = tg[(r0 + rr) * LW + xl + b];
        if (ii >= 0 && ii < H) {
#pragma unroll
          for (int a = 0; a < K; ++a) {
            const int o = rr - a;
            if (o >= 0 && o < RPT) {
#pragma unroll
              for (int b = 0; b < K; ++b) acc[o] += wr[(K - 1 - a) * K + (K - 1 - b)] * val[b];
            }
          }
        } else {
#pragma unroll
          for (int a = 0; a < K; ++a) {
            const int o = rr - a;
            if (o >= 0 && o < RPT) {
              const int yy = tp.ty0 + r0 + o;
              const bool feeds = (ii < 0) ? (yy >= 1) : (yy <= H - 2);
              if (feeds) {
#pragma unroll
                for (int b = 0; b < K; ++b) acc[o] += wr[a * K + (K - 1 - b)] * val[b];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const ubase_t gp = uniform_base(gx + off);
#pragma unroll
      for (int o = 0; o < RPT; ++o) {
        const int yy = tp.ty0 + r0 + o;
        float extra = 0.f;
        const bool south = yy >= 1 && yy <= P, north = yy >= H - 1 - P && yy <= H - 2;
        if (south || north) {                   // (wave-uniform; the W/2-shifted pole row comes from memory)
          const int a = south ? P - yy : P + (H - 1 - yy);
          const int prow = south ? 0 : H - 1;
#pragma unroll
          for (int b = 0; b < K; ++b) {
            int col = xx + P - b + half;
            if (col >= W) col -= W;
            if (col >= W) col -= W;
            extra += wc[a * K + b] * gpl[(int64_t)prow * W + col];
          }
        }
        if (col_owned && o >= own0)
          store_at<float>(gp + (int64_t)o * W * 4, o0, ADD ? (acc[o] + extra) + av[o] : acc[o] + extra);
      }
    }
    // ---- weight gradient: the owned points' share of the channel's sums
    if constexpr (WG) {
      float g[RPT];
#pragma unroll
      for (int o = 0; o < RPT; ++o) {
        g[o] = (col_owned && o >= own0) ? tg[(r0 + o + P) * LW + xl + P] : 0.f;
        gsum += g[o];
      }
#pragma unroll
      for (int rr = 0; rr < RPT + K - 1; ++rr) {
        float val[K];
#pragma unroll
        for (int b = 0; b < K; ++b) val[b] = tx[(r0 + rr) * LW + xl + b];
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const int o = rr - a;
          if (o >= 0 && o < RPT) {
#pragma unroll
            for (int b = 0; b < K; ++b) accw[a * K + b] += g[o] * val[b];
          }
        }
      }
    }
    __syncthreads();
  }
  if (!WG) return;
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    float s = wave_sum_dpp(accw[i]);
    if (xl == 0) red[wave][i] = s;
  }
  {
    float s = wave_sum_dpp(gsum);
    if (xl == 0) red[wave][K * K] = s;
  }
  __syncthreads();
  if (threadIdx.x < NW) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[((int64_t)c * chunks + chunk) * NW + threadIdx.x] = s;
  }
}

__global__ void __launch_bounds__(256)
dwconv_wgrad_finish(const float* __restrict__ partial, float* __restrict__ gw,
                    float* __restrict__ gbias, int C, int KK, int chunks) {
  const int NW = KK + 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= C * NW) return;
  const int c = idx / NW, i = idx - c * NW;
  float s = 0.f;
  for (int ch = 0; ch < chunks; ++ch) s += partial[((int64_t)c * chunks + ch) * NW + i];
  if (i < KK) gw[(int64_t)c * KK + i] = s;
  else if (gbias) gbias[c] = s;
}

// ---------------------------------------------------------------------------- avgpool
__global__ void __launch_bounds__(256)
avgpool_geo_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t planes, int H,
                       int W, int Ho, int Wo, int s) {
  const int64_t per = (int64_t)Ho * Wo, total = planes * per;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t plane = idx / per;
    const int rem = (int)(idx - plane * per);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float* xp = x + plane * (int64_t)H * W;
    float sum = 0.f;
    for (int a = 0; a < 5; ++a)
      for (int b = 0; b < 5; ++b) {
        int r, c;
        geo_src(oy * s + a - 2, ox * s + b - 2, H, W, r, c);
        sum += xp[(int64_t)r * W + c];
      }
    y[idx] = sum / 25.0f;
  }
}

__global__ void __launch_bounds__(256)
avgpool_geo_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes, int H,
                       int W, int Ho, int Wo, int s) {
  const int64_t per = (int64_t)H * W, total = planes * per;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t plane = idx / per;
    const int rem = (int)(idx - plane * per);
    const int yy = rem / W, xx = rem - yy * W;
    const float* g = gy + plane * (int64_t)Ho * Wo;
    float acc = 0.f;
    geo_for_each_alias(yy, xx, H, W, 2, [&](int ii, int jj) {
      const int r = ii + 2, c = jj + 2;  // padded coordinates
      // windows [o*s, o*s+4] covering r / c
      int oy_lo = (r - 4 + s - 1) / s; if (r - 4 < 0) oy_lo = 0;
      int ox_lo = (c - 4 + s - 1) / s; if (c - 4 < 0) ox_lo = 0;
      const int oy_hi = min(r / s, Ho - 1), ox_hi = min(c / s, Wo - 1);
      for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) acc += g[(int64_t)oy * Wo + ox];
    });
    gx[idx] = acc / 25.0f;
  }
}

// ---------------------------------------------------------------------------- upsample
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_index(int o, int in_size, int out_size) {
  Lerp L;
  if (in_size == out_size) { L.i0 = L.i1 = o; L.l0 = 1.f; L.l1 = 0.f; return L; }
  const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float real = scale * (float)o;
  L.i0 = (int)real;
  L.i1 = L.i0 + ((L.i0 < in_size - 1) ? 1 : 0);
  L.l1 = fminf(fmaxf(real - (float)L.i0, 0.f), 1.f);
  L.l0 = 1.f - L.l1;
  return L;
}

__global__ void __launch_bounds__(256)
upsample_lonp_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes, int Hc,
                     int Wc, int H, int W) {
  // src = coarse x, dst = fine y   (the adjoint is the gather kernel below)
  const int64_t per = (int64_t)H * W, total = planes * per;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t plane = idx / per;
    const int rem = (int)(idx - plane * per);
    const int h = rem / W, w = rem - h * W;
    const Lerp lh = lerp_index(h, Hc, H);
    const Lerp lw = lerp_index(w, Wc + 1, W + 1);  // periodic column appended on both sides
    const int c0 = lw.i0 >= Wc ? lw.i0 - Wc : lw.i0, c1 = lw.i1 >= Wc ? lw.i1 - Wc : lw.i1;
    const int64_t base = plane * (int64_t)Hc * Wc;
    const float* xp = src + base;
    const float top = lw.l0 * xp[(int64_t)lh.i0 * Wc + c0] + lw.l1 * xp[(int64_t)lh.i0 * Wc + c1];
    const float bot = lw.l0 * xp[(int64_t)lh.i1 * Wc + c0] + lw.l1 * xp[(int64_t)lh.i1 * Wc + c1];
    dst[idx] = lh.l0 * top + lh.l1 * bot;
  }
}

// Adjoint of the upsampling as a GATHER (round 4; rounds 1-3 scattered four float atomics per fine point): a thread
// owns one coarse cell and walks the fine points that can reference it - a conservative index range per axis, each
// candidate re-evaluated with the forward's own lerp_index, so no inverse of the float index map is needed.  No
// atomics, no zero fill, one fixed summation order: bitwise reproducible.
__global__ void __launch_bounds__(256)
upsample_lonp_bwd_gather_kernel(const float* __restrict__ gy, float* __restrict__ gx, int64_t planes, int Hc, int Wc,
                                int H, int W) {
  const int64_t per = (int64_t)Hc * Wc, total = planes * per;
  const float inv_h = Hc > 1 ? (float)(H - 1) / (float)(Hc - 1) : 0.f;     // fine rows per coarse row
  const float inv_w = (float)W / (float)Wc;                                  // fine columns per coarse column
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t plane = idx / per;
    const int rem = (int)(idx - plane * per);
    const int hc = rem / Wc, wc = rem - hc * Wc;
    const float* g = gy + plane * (int64_t)H * W;
    const int h_lo = Hc > 1 ? max(0, (int)floorf((float)(hc - 1) * inv_h) - 1) : 0;
    const int h_hi = Hc > 1 ? min(H - 1, (int)ceilf((float)(hc + 1) * inv_h) + 1) : H - 1;
    float acc = 0.f;
    for (int h = h_lo; h <= h_hi; ++h) {
      const Lerp lh = lerp_index(h, Hc, H);
      const float wh = (lh.i0 == hc ? lh.l0 : 0.f) + (lh.i1 == hc ? lh.l1 : 0.f);
      if (wh == 0.f) continue;
      float rowacc = 0.f;
      // two candidate ranges of fine columns: around the coarse column, and - column 0 only - the end of the
      // circle, whose right neighbour is the appended periodic column
      for (int part = 0; part < 2; ++part) {
        int w_lo, w_hi;
        if (part == 0) {
          w_lo = max(0, (int)floorf((float)(wc - 1) * inv_w) - 1);
          w_hi = min(W - 1, (int)ceilf((float)(wc + 1) * inv_w) + 1);
        } else {
          if (wc != 0) break;
          w_lo = max(0, (int)floorf((float)(Wc - 1) * inv_w) - 1);
          w_hi = W - 1;
          const int first_hi = min(W - 1, (int)ceilf(inv_w) + 1);      // (do not visit a column twice)
          w_lo = max(w_lo, first_hi + 1);
        }
        for (int w = w_lo; w <= w_hi; ++w) {
          const Lerp lw = lerp_index(w, Wc + 1, W + 1);
          const int c0 = lw.i0 >= Wc ? lw.i0 - Wc : lw.i0, c1 = lw.i1 >= Wc ? lw.i1 - Wc : lw.i1;
          const float ww = (c0 == wc ? lw.l0 : 0.f) + (c1 == wc ? lw.l1 : 0.f);
          if (ww != 0.f) rowacc = fmaf(g[(int64_t)h * W + w], ww, rowacc);
        }
      }
      acc = fmaf(rowacc, wh, acc);
    }
    gx[idx] = acc;
  }
}

int check_dw(const char* name, int B, int C, int H, int W, int k) {
  PD_REQUIRE(B >= 0 && C >= 1 && H >= 2 && W >= 2, "%s: bad shape", name);
  PD_REQUIRE(k >= 1 && k <= 11 && (k & 1), "%s: kernel size %d not supported (odd sizes 1..11)", name, k);
  PD_REQUIRE(W % 2 == 0, "%s: Number of longitude points must be even", name);
  PD_REQUIRE((k - 1) / 2 <= H - 2 && (k - 1) <= W, "%s: grid %dx%d too small for k=%d", name, H, W, k);
  // the analytic halo folding of the data gradient is verified for halos that do not overlap themselves
  PD_REQUIRE(k <= 7 || (H >= 2 * k && W >= 2 * k), "%s: k=%d needs a grid of at least %dx%d", name, k, 2 * k, 2 * k);
  PD_REQUIRE((int64_t)B * C * (((H + TH - 1) / TH) * ((W + TW - 1) / TW)) < (1ll << 31), "%s: too large", name);
  return 0;
}

int wgrad_chunks(int B, int C, int tiles) {
  int items = B * tiles;
  int chunks = (2048 + C - 1) / C;
  return std::max(1, std::min(chunks, items));
}

// staged-tiles backward: items per workgroup and workgroups per channel (~8192 workgroups: 8 resident sets of the chip)
int bwd_tiles_per(int B, int C, int tiles) {
  const int items = std::max(1, B * tiles);
  const int chunks = std::max(1, std::min((8192 + C - 1) / C, items));
  return (items + chunks - 1) / chunks;
}
int bwd_tiles_chunks(int B, int C, int tiles) {
  const int items = std::max(1, B * tiles), per = bwd_tiles_per(B, C, tiles);
  return (items + per - 1) / per;
}

}  // namespace

// whole padded plane == one tile and the 16-byte staging path applies (even halo: k = 5)
static int whole_plane_vec4(const float* src, int H, int W, int k) {
  return k == 5 && W == TW && H <= TH && ((int64_t)H * W) % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(src) & 15) == 0;
}

#ifndef DWCONV_TILES             // (A/B builds: 0 = the one-tile-per-workgroup kernels on every larger grid)
#define DWCONV_TILES 1
#endif
// more than one tile, and the staged full-tile kernels apply: k = 5, at least one full tile each way, rows of whole
// float4, 16-byte aligned tensors (every plane then is), byte offsets in a plane fit 32 bits
static bool staged_tiles(const void* a, const void* b, int H, int W, int k) {
  return DWCONV_TILES && k == 5 && H >= TH && W >= TW && (H > TH || W > TW) && W % 4 == 0 &&
         (int64_t)H * W * 4 < (1ll << 32) && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

#define DISPATCH_K(k, CALL)          \
  switch (k) {                       \
    case 1: { constexpr int KK = 1; CALL; } break; \
    case 3: { constexpr int KK = 3; CALL; } break; \
    case 5: { constexpr int KK = 5; CALL; } break; \
    case 7: { constexpr int KK = 7; CALL; } break; \
    case 9: { constexpr int KK = 9; CALL; } break; \
    default: { constexpr int KK = 11; CALL; } break; \
  }

template <bool Y16>
static int dwconv_geo_fwd_impl(const float* x, const float* w, const float* bias, float* y,
                               int B, int C, int H, int W, int k, void* stream) {
  if (int e = check_dw("dwconv_geo_fwd", B, C, H, W, k)) return e;
  if (B == 0) return 0;
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH, tiles = tx * ty;
  const int64_t planes = (int64_t)B * C;
  if (DWCONV_PLANES && whole_plane_vec4(x, H, W, k)) {     // (k == 5 there)
    hipLaunchKernelGGL((dwconv_geo_fwd_planes_kernel<5, Y16>), dim3((unsigned)((planes + PLANE_CHUNK - 1) / PLANE_CHUNK)),
                       dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, H, planes);
    PD_CHECK_LAUNCH("dwconv_geo_fwd");
    return 0;
  }
  if (staged_tiles(x, y, H, W, k)) {
    const int64_t nwg = (planes + PLANE_CHUNK - 1) / PLANE_CHUNK * tiles;
    hipLaunchKernelGGL((dwconv_geo_fwd_tiles_kernel<5, Y16>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       y, C, H, W, tx, tiles, planes);
    PD_CHECK_LAUNCH("dwconv_geo_fwd");
    return 0;
  }
  const unsigned grid = (unsigned)(planes * tiles);
  DISPATCH_K(k, hipLaunchKernelGGL((dwconv_geo_fwd_kernel<KK, Y16>), dim3(grid), dim3(256), 0,
                                   (hipStream_t)stream, x, w, bias, y, C, H, W, tx, tiles,
                                   whole_plane_vec4(x, H, W, k)));
  PD_CHECK_LAUNCH("dwconv_geo_fwd");
  return 0;
}

extern "C" int paradis_dwconv_geo_fwd(const float* x, const float* w, const float* bias, float* y,
                                      int B, int C, int H, int W, int k, void* stream) {
  return dwconv_geo_fwd_impl<false>(x, w, bias, y, B, C, H, W, k, stream);
}

// y written as bf16 (ABI 9; bf16-mixed mode: the consumer is the SepConv's pointwise GEMM)
extern "C" int paradis_dwconv_geo_fwd16(const float* x, const float* w, const float* bias, void* y,
                                        int B, int C, int H, int W, int k, void* stream) {
  return dwconv_geo_fwd_impl<true>(x, w, bias, (float*)y, B, C, H, W, k, stream);
}

static int dwconv_geo_dgrad_launch(const float* gy, const float* w, const float* addend, float* gx, int B, int C,
                                   int H, int W, int k, void* stream) {
  if (int e = check_dw("dwconv_geo_dgrad", B, C, H, W, k)) return e;
  if (B == 0) return 0;
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH, tiles = tx * ty;
  if (DWCONV_PLANES && whole_plane_vec4(gy, H, W, k) && (reinterpret_cast<uintptr_t>(gx) & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(addend) & 3) == 0) {
    const int64_t planes = (int64_t)B * C;
    const dim3 grid((unsigned)((planes + PLANE_CHUNK - 1) / PLANE_CHUNK));
    if (addend)
      hipLaunchKernelGGL((dwconv_geo_dgrad_planes_kernel<5, true>), grid, dim3(256), 0, (hipStream_t)stream, gy, w,
                         addend, gx, C, H, planes);
    else
      hipLaunchKernelGGL((dwconv_geo_dgrad_planes_kernel<5, false>), grid, dim3(256), 0, (hipStream_t)stream, gy, w,
                         addend, gx, C, H, planes);
    PD_CHECK_LAUNCH("dwconv_geo_dgrad");
    return 0;
  }
  if (staged_tiles(gy, gx, H, W, k) && staged_tiles(gy, addend, H, W, k)) {     // the one-pass kernel's data-gradient half
    const int per = bwd_tiles_per(B, C, tiles), chunks = bwd_tiles_chunks(B, C, tiles);
    const float* none = nullptr;
    float* nopart = nullptr;
    if (addend)
      hipLaunchKernelGGL((dwconv_geo_bwd_tiles_kernel<5, true, true, false>), dim3(C * chunks), dim3(256), 0,
                         (hipStream_t)stream, gy, none, w, addend, gx, nopart, B, C, H, W, tx, tiles, chunks, per);
    else
      hipLaunchKernelGGL((dwconv_geo_bwd_tiles_kernel<5, false, true, false>), dim3(C * chunks), dim3(256), 0,
                         (hipStream_t)stream, gy, none, w, addend, gx, nopart, B, C, H, W, tx, tiles, chunks, per);
    PD_CHECK_LAUNCH("dwconv_geo_dgrad");
    return 0;
  }
  const unsigned grid = (unsigned)((int64_t)B * C * tiles);
  DISPATCH_K(k, hipLaunchKernelGGL(dwconv_geo_dgrad_kernel<KK>, dim3(grid), dim3(256), 0,
                                   (hipStream_t)stream, gy, w, addend, gx, C, H, W, tx, tiles,
                                   whole_plane_vec4(gy, H, W, k)));
  PD_CHECK_LAUNCH("dwconv_geo_dgrad");
  return 0;
}

extern "C" int paradis_dwconv_geo_dgrad(const float* gy, const float* w, float* gx, int B, int C,
                                        int H, int W, int k, void* stream) {
  return dwconv_geo_dgrad_launch(gy, w, nullptr, gx, B, C, H, W, k, stream);
}

// gx = dgrad(gy) + addend (addend [B,C,H,W], not aliasing gx)
extern "C" int paradis_dwconv_geo_dgrad_add(const float* gy, const float* w, const float* addend, float* gx, int B,
                                            int C, int H, int W, int k, void* stream) {
  PD_REQUIRE(B == 0 || (addend != nullptr && addend != gx), "dwconv_geo_dgrad_add: addend must be a tensor other than gx");
  return dwconv_geo_dgrad_launch(gy, w, addend, gx, B, C, H, W, k, stream);
}

extern "C" size_t paradis_dwconv_geo_wgrad_ws_bytes(int B, int C, int H, int W, int k) {
  const int tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  const int chunks = std::max(wgrad_chunks(B, C, tiles), bwd_tiles_chunks(B, C, tiles));
  return (size_t)C * chunks * (k * k + 1) * sizeof(float) + 256;
}

extern "C" int paradis_dwconv_geo_wgrad(const float* gy, const float* x, float* gw, float* gbias,
                                        int B, int C, int H, int W, int k, void* workspace,
                                        void* stream) {
  if (int e = check_dw("dwconv_geo_wgrad", B, C, H, W, k)) return e;
  PD_REQUIRE(workspace != nullptr, "dwconv_geo_wgrad: workspace required");
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH, tiles = tx * ty;
  const bool staged = B > 0 && staged_tiles(gy, x, H, W, k);       // the one-pass kernel's weight-gradient half
  const int chunks = B == 0 ? 1 : (staged ? bwd_tiles_chunks(B, C, tiles) : wgrad_chunks(B, C, tiles));
  float* partial = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (staged) {
    const float* none = nullptr;
    float* nogx = nullptr;
    hipLaunchKernelGGL((dwconv_geo_bwd_tiles_kernel<5, false, false, true>), dim3(C * chunks), dim3(256), 0, st, gy, x, none,
                       none, nogx, partial, B, C, H, W, tx, tiles, chunks, bwd_tiles_per(B, C, tiles));
  } else if (DWCONV_PLANES && whole_plane_vec4(x, H, W, k) && (reinterpret_cast<uintptr_t>(gy) & 3) == 0)
    hipLaunchKernelGGL(dwconv_geo_wgrad_planes_kernel<5>, dim3(C * chunks), dim3(256), 0, st, gy, x, partial, B, C,
                       H, chunks);
  else
    DISPATCH_K(k, hipLaunchKernelGGL(dwconv_geo_wgrad_kernel<KK>, dim3(C * chunks), dim3(256), 0, st, gy,
                                     x, partial, B, C, H, W, tx, tiles, chunks, whole_plane_vec4(x, H, W, k)));
  const int n = C * (k * k + 1);
  hipLaunchKernelGGL(dwconv_wgrad_finish, dim3((n + 255) / 256), dim3(256), 0, st, partial, gw, gbias,
                     C, k * k, chunks);
  PD_CHECK_LAUNCH("dwconv_geo_wgrad");
  return 0;
}

template <bool GY16>
static int dwconv_geo_bwd_planes_launch(const float* gy, const float* x, const float* w, const float* addend, float* gx,
                                        float* gw, float* gbias, int B, int C, int H, int k, void* workspace, void* stream) {
  const int chunks = wgrad_chunks(B, C, 1);
  float* partial = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (addend)
    hipLaunchKernelGGL((dwconv_geo_bwd_planes_kernel<5, true, GY16>), dim3(C * chunks), dim3(256), 0, st, gy, x, w, addend,
                       gx, partial, B, C, H, chunks);
  else
    hipLaunchKernelGGL((dwconv_geo_bwd_planes_kernel<5, false, GY16>), dim3(C * chunks), dim3(256), 0, st, gy, x, w, addend,
                       gx, partial, B, C, H, chunks);
  const int n = C * (k * k + 1);
  hipLaunchKernelGGL(dwconv_wgrad_finish, dim3((n + 255) / 256), dim3(256), 0, st, partial, gw, gbias, C, k * k,
                     chunks);
  PD_CHECK_LAUNCH("dwconv_geo_bwd");
  return 0;
}

// Both gradients of the stencil from one call: gx = dgrad(gy) (+ addend), gw / gbias.  On the whole-plane path
// (k = 5, W = 64, H <= 32, aligned tensors: the reference grids at 5.625 degrees) and on the staged-tiles path (k = 5,
// larger grids with W % 4 == 0) ONE kernel reads gy once; elsewhere the two kernels of paradis_dwconv_geo_dgrad / _wgrad
// run one after the other.  Bit-identical to paradis_dwconv_geo_dgrad(_add) and paradis_dwconv_geo_wgrad on every path.
// workspace: paradis_dwconv_geo_wgrad_ws_bytes.  addend, gbias: nullable.
extern "C" int paradis_dwconv_geo_bwd(const float* gy, const float* x, const float* w, const float* addend, float* gx,
                                      float* gw, float* gbias, int B, int C, int H, int W, int k, void* workspace,
                                      void* stream) {
  if (int e = check_dw("dwconv_geo_bwd", B, C, H, W, k)) return e;
  PD_REQUIRE(workspace != nullptr, "dwconv_geo_bwd: workspace required");
  PD_REQUIRE(addend == nullptr || addend != gx, "dwconv_geo_bwd: addend must not alias gx");
  const bool fused = DWCONV_BWD_FUSED && DWCONV_PLANES && B > 0 && whole_plane_vec4(gy, H, W, k) &&
                     whole_plane_vec4(x, H, W, k) && (reinterpret_cast<uintptr_t>(gx) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(addend) & 3) == 0;
  if (B > 0 && staged_tiles(gy, x, H, W, k) && staged_tiles(gx, addend, H, W, k)) {
    const int tx = (W + TW - 1) / TW, tiles = tx * ((H + TH - 1) / TH);
    const int per = bwd_tiles_per(B, C, tiles), chunks = bwd_tiles_chunks(B, C, tiles);
    float* partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (addend)
      hipLaunchKernelGGL((dwconv_geo_bwd_tiles_kernel<5, true, true, true>), dim3(C * chunks), dim3(256), 0, st, gy, x, w,
                         addend, gx, partial, B, C, H, W, tx, tiles, chunks, per);
    else
      hipLaunchKernelGGL((dwconv_geo_bwd_tiles_kernel<5, false, true, true>), dim3(C * chunks), dim3(256), 0, st, gy, x, w,
                         addend, gx, partial, B, C, H, W, tx, tiles, chunks, per);
    const int n = C * (k * k + 1);
    hipLaunchKernelGGL(dwconv_wgrad_finish, dim3((n + 255) / 256), dim3(256), 0, st, partial, gw, gbias, C, k * k, chunks);
    PD_CHECK_LAUNCH("dwconv_geo_bwd");
    return 0;
  }
  if (!fused) {
    // (what is left here: k != 5, rows that are not whole float4, grids smaller than a tile one way.  A first one-pass
    //  kernel for the larger grids - one channel per workgroup, (sample, tile) items strided over two workgroups, both
    //  tiles staged synchronously per item - measured 1004 us per call at 128 x 256, B = 8, C = 1024 against 628 + 483 us
    //  for these two kernels; dwconv_geo_bwd_tiles_kernel above - next item's loads in flight, tile-fastest item order,
    //  ~8192 workgroups - takes 822 us)
    if (int e = dwconv_geo_dgrad_launch(gy, w, addend, gx, B, C, H, W, k, stream)) return e;
    return paradis_dwconv_geo_wgrad(gy, x, gw, gbias, B, C, H, W, k, workspace, stream);
  }
  return dwconv_geo_bwd_planes_launch<false>(gy, x, w, addend, gx, gw, gbias, B, C, H, k, workspace, stream);
}

// gy as a bf16 tensor (ABI 9; bf16-mixed mode), everything else as paradis_dwconv_geo_bwd.  Whole-plane grids only
// (paradis_dwconv_geo_bwd16_ok: k = 5, W = 64, H <= 32 - the 5.625-degree grid); the caller widens gy elsewhere.
extern "C" int paradis_dwconv_geo_bwd16_ok(int H, int W, int k) {
  return (DWCONV_BWD_FUSED && DWCONV_PLANES && k == 5 && W == TW && H <= TH && ((int64_t)H * W) % 4 == 0) ? 1 : 0;
}
extern "C" int paradis_dwconv_geo_bwd16(const void* gy, const float* x, const float* w, const float* addend, float* gx,
                                        float* gw, float* gbias, int B, int C, int H, int W, int k, void* workspace,
                                        void* stream) {
  if (int e = check_dw("dwconv_geo_bwd16", B, C, H, W, k)) return e;
  PD_REQUIRE(workspace != nullptr, "dwconv_geo_bwd16: workspace required");
  PD_REQUIRE(addend == nullptr || addend != gx, "dwconv_geo_bwd16: addend must not alias gx");
  PD_REQUIRE(paradis_dwconv_geo_bwd16_ok(H, W, k), "dwconv_geo_bwd16: whole-plane grids only (%dx%d, k = %d)", H, W, k);
  PD_REQUIRE(((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
             ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(addend)) & 3) == 0,
             "dwconv_geo_bwd16: misaligned tensor");
  if (B == 0) return paradis_dwconv_geo_bwd(nullptr, x, w, addend, gx, gw, gbias, B, C, H, W, k, workspace, stream);
  return dwconv_geo_bwd_planes_launch<true>((const float*)gy, x, w, addend, gx, gw, gbias, B, C, H, k, workspace, stream);
}

static int check_pool(const char* name, int64_t planes, int H, int W, int s) {
  PD_REQUIRE(planes >= 0 && H >= 4 && W >= 4 && W % 2 == 0, "%s: bad shape %dx%d", name, H, W);
  PD_REQUIRE(s >= 1, "%s: Coarsening factor must be >=1", name);
  return 0;
}

extern "C" int paradis_avgpool_geo_fwd(const float* x, float* y, int64_t planes, int H, int W,
                                       int stride, void* stream) {
  if (int e = check_pool("avgpool_geo_fwd", planes, H, W, stride)) return e;
  if (planes == 0) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t total = planes * Ho * Wo;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
  hipLaunchKernelGGL(avgpool_geo_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y,
                     planes, H, W, Ho, Wo, stride);
  PD_CHECK_LAUNCH("avgpool_geo_fwd");
  return 0;
}

extern "C" int paradis_avgpool_geo_bwd(const float* gy, float* gx, int64_t planes, int H, int W,
                                       int stride, void* stream) {
  if (int e = check_pool("avgpool_geo_bwd", planes, H, W, stride)) return e;
  if (planes == 0) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int64_t total = planes * H * W;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
  hipLaunchKernelGGL(avgpool_geo_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gy, gx,
                     planes, H, W, Ho, Wo, stride);
  PD_CHECK_LAUNCH("avgpool_geo_bwd");
  return 0;
}

extern "C" int paradis_upsample_lonp_fwd(const float* x, float* y, int64_t planes, int Hc, int Wc,
                                         int H, int W, void* stream) {
  PD_REQUIRE(planes >= 0 && Hc >= 1 && Wc >= 1 && H >= Hc && W >= Wc, "upsample_lonp_fwd: bad shape");
  if (planes == 0) return 0;
  const int64_t total = planes * H * W;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
  hipLaunchKernelGGL(upsample_lonp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y,
                     planes, Hc, Wc, H, W);
  PD_CHECK_LAUNCH("upsample_lonp_fwd");
  return 0;
}

extern "C" int paradis_upsample_lonp_bwd(const float* gy, float* gx, int64_t planes, int Hc, int Wc,
                                         int H, int W, void* stream) {
  PD_REQUIRE(planes >= 0 && Hc >= 1 && Wc >= 1 && H >= Hc && W >= Wc, "upsample_lonp_bwd: bad shape");
  if (planes == 0) return 0;
  const int64_t total = planes * Hc * Wc;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
  hipLaunchKernelGGL(upsample_lonp_bwd_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gy, gx,
                     planes, Hc, Wc, H, W);
  PD_CHECK_LAUNCH("upsample_lonp_bwd");
  return 0;
}
