// a3-a5: fused semi-Lagrangian advection core (reference model/advection.py:129-169).
//
//   F~  = F with rows 0, H-1 replaced by their longitudinal mean       (advection.py:100-114)
//   (phi, lam) departure point in the rotated frame                     (advection.py:74-98)
//   sample coordinates on the geocyclic-padded plane                    (advection.py:139-150 +
//       ATen grid_sampler unnormalise, align_corners=True)
//   bilinear / bicubic (Keys A=-0.75) gather, rows 0, H-1 of the result replaced by their mean
//
// Saved-for-backward state is (F, u, v) only; everything else is recomputed.
// Algorithmic HBM traffic: 16 B/point forward, 28 B/point backward.
//
// Design (round 1, second version).  The kernels are VALU-bound, not HBM-bound, so the per-point
// instruction count is what matters.  A workgroup stages a WINDOW of the geocyclic-PADDED plane in
// LDS through the a1 index map (pole rows already replaced by their means); the 4x4 / 2x2 taps of a
// point are then plain 2-D indexing from one base address - no per-tap wrap / mirror / validity
// logic.  Two instantiations of the same kernel:
//   WHOLE : window = the whole padded plane (planes up to 32x64..64x64), one workgroup per plane,
//           pole means computed in LDS, no fallback path;
//   tiled : 16x128 arrival tiles with a halo of D padded cells; taps outside the window (large
//           displacement, over-the-pole flow) fall back to global loads through the index map /
//           global float atomics; pole rows by two tiny pre/post kernels.
// Backward scatters into 64-bit fixed-point window accumulators with integer LDS atomics
// (ds_add_f32 is ~30x slower than ds_add_u64 on gfx950, tools/lds_atomic_bench.hip), scaled by an
// exact power of two from the tile's max |cotangent| (resolution max|g| 2^-41), then folds the halo
// back (WHOLE) or flushes the window once with global atomics (tiled).
//
// The coordinate chain keeps the reference's fp32 operation order; this file is compiled with
// -ffp-contract=off (see Makefile) so that no extra FMAs are formed (explicit fmaf is used only
// inside the sincos range reduction).
#include <stdlib.h>
#include <algorithm>
#include "common.h"

#pragma clang fp contract(off)

// Diagnostic ablation switches (tools/advect_variants.py builds side libraries with them; the
// shipped library defines none):  ADV_NO_ATOMIC, ADV_NO_TRIG, ADV_OCML_SINCOS
namespace {

constexpr float TWO_PI_F = 6.283185307179586f;
constexpr float CLAMP_HI = 0.9999999f;  // float(1 - 1e-7), as torch.clamp converts its python bound
constexpr float KA = -0.75f;
constexpr int TILE_H = 16, TILE_W = 128;  // arrival tile of the tiled schedule
#ifndef ADV_UNROLL
#define ADV_UNROLL 4
#endif
#ifndef ADV_UNROLL_BWD
#define ADV_UNROLL_BWD 2
#endif

struct AdvGeom {
  int H, W, p;
  float dt, min_lat, min_lon, d_lat, d_lon;
};

struct DepState {  // intermediates needed by the backward chain
  float sp, cp, sl, cl, s, n, d;
};

// sin and cos with a Cody-Waite reduction (fdlibm's float split of pi/2) and the cephes minimax
// polynomials on [-pi/4, pi/4] (<= ~1 ulp); huge arguments take the ocml path.
__device__ __forceinline__ void sincos_fast(float x, float& s, float& c) {
#ifdef ADV_OCML_SINCOS
  sincosf(x, &s, &c);
#else
  if (fabsf(x) > 8192.0f) {
    sincosf(x, &s, &c);
    return;
  }
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5707855225e+00f, x);   // pi/2 split in three parts with trailing zero bits
  r = fmaf(-k, 1.0804273188e-05f, r);
  r = fmaf(-k, 6.0770999344e-11f, r);
  const int q = (int)k;
  const float z = r * r;
  const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
                   0.5f * z + 1.0f;
  const float ss = (q & 1) ? pc : ps;
  const float cc = (q & 1) ? ps : pc;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
#endif
}

// exact fmod(t, 2*pi_f) for the range the path produces (Sterbenz: the subtractions are exact);
// anything else takes fmodf + the sign fix of torch.remainder
__device__ __forceinline__ float wrap_two_pi(float t) {
  if (t >= 0.f && t < 3.0f * TWO_PI_F) {
    if (t >= 2.0f * TWO_PI_F) return t - 2.0f * TWO_PI_F;
    if (t >= TWO_PI_F) return t - TWO_PI_F;
    return t;
  }
  float m = fmodf(t, TWO_PI_F);
  if (m != 0.f && m < 0.f) m += TWO_PI_F;
  return m;
}

__device__ __forceinline__ void departure(float u, float v, float sa, float ca, float lon_a,
                                          const AdvGeom& g, float& ix, float& iy, DepState* st) {
#ifdef ADV_NO_TRIG
  ix = lon_a * 3.0f + u + (float)g.p; iy = sa * 5.0f + 7.0f + v + (float)g.p;
  if (st) { st->sp = u; st->cp = v; st->sl = sa; st->cl = ca; st->s = 0.5f; st->n = u; st->d = 1.0f + v * v; }
  return;
#endif
  const float lam = -u * g.dt;
  const float phi = -v * g.dt;
  float sp, cp, sl, cl;
  sincos_fast(phi, sp, cp);
  sincos_fast(lam, sl, cl);
  const float cc = cp * cl;
  const float s = sp * ca + cc * sa;
  const float sc = fminf(fmaxf(s, -CLAMP_HI), CLAMP_HI);
  const float lat_d = asinf(sc);
  const float n = cp * sl;
  const float d = cc * ca - sp * sa;
  float lon_d = lon_a + atan2f(n, d);
  lon_d = lon_d + TWO_PI_F;
  const float m = wrap_two_pi(lon_d);
  const float pix_x = (m - g.min_lon) / g.d_lon * ((float)g.W - 1.0f);
  const float pix_y = (lat_d - g.min_lat) / g.d_lat * ((float)g.H - 1.0f);
  const float wpm1 = (float)(g.W + 2 * g.p - 1), hpm1 = (float)(g.H + 2 * g.p - 1);
  const float gx = 2.0f * ((pix_x + (float)g.p) / wpm1) - 1.0f;
  const float gy = 2.0f * ((pix_y + (float)g.p) / hpm1) - 1.0f;
  ix = ((gx + 1.0f) / 2.0f) * wpm1;
  iy = ((gy + 1.0f) / 2.0f) * hpm1;
  if (st) {
    st->sp = sp; st->cp = cp; st->sl = sl; st->cl = cl; st->s = s; st->n = n; st->d = d;
  }
}

__device__ __forceinline__ float cub1(float x) { return ((KA + 2.f) * x - (KA + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cub2(float x) { return ((KA * x - 5.f * KA) * x + 8.f * KA) * x - 4.f * KA; }
__device__ __forceinline__ float dcub1(float x) { return (3.f * (KA + 2.f) * x - 2.f * (KA + 3.f)) * x; }
__device__ __forceinline__ float dcub2(float x) { return (3.f * KA * x - 10.f * KA) * x + 8.f * KA; }

template <int MODE>
struct Interp {
  static constexpr int NT = (MODE == PARADIS_INTERP_BICUBIC) ? 4 : 2;
  static constexpr int OFF0 = (MODE == PARADIS_INTERP_BICUBIC) ? -1 : 0;
  static __device__ __forceinline__ void weights(float t, float* w) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      w[0] = cub2(t + 1.f); w[1] = cub1(t); w[2] = cub1(1.f - t); w[3] = cub2(2.f - t);
    } else {
      w[0] = 1.f - t; w[1] = t;
    }
  }
  static __device__ __forceinline__ void dweights(float t, float* dw) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      dw[0] = dcub2(t + 1.f); dw[1] = dcub1(t); dw[2] = -dcub1(1.f - t); dw[3] = -dcub2(2.f - t);
    } else {
      dw[0] = -1.f; dw[1] = 1.f;
    }
  }
};

// ATen zeroes taps outside the padded plane; on this path that only happens when a coordinate
// rounds onto the plane edge, where the outside taps carry weight exactly 0.  The tap block is
// therefore shifted inside the plane (by `shift` cells) and the weights re-indexed: cells that left
// the block get weight 0 - same value, no stray reads.
template <int NT>
__device__ __forceinline__ void shift_weights(float* w, int shift) {
  if (shift == 0) return;
  float t[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) t[b] = w[b];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k) v = (k == b + shift) ? t[k] : v;
    w[b] = v;
  }
}

// Top-left tap (padded coordinates), clamped inside the plane; sx/sy = applied shifts.
template <int MODE>
__device__ __forceinline__ void tap_origin(float ix, float iy, int Hp, int Wp, int& bx, int& by,
                                           int& sx, int& sy, float& tx, float& ty) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  const float x0f = floorf(ix), y0f = floorf(iy);
  tx = ix - x0f;
  ty = iy - y0f;
  const bool sane = (fabsf(ix) < 1e8f) && (fabsf(iy) < 1e8f);  // NaN/inf: every tap lies outside
  const int x0 = sane ? (int)x0f + OFF0 : -1000000, y0 = sane ? (int)y0f + OFF0 : -1000000;
  bx = min(max(x0, 0), Wp - NT);
  by = min(max(y0, 0), Hp - NT);
  sx = bx - x0;
  sy = by - y0;
}

__device__ __forceinline__ float wave_row_mean(const float* row, int W) {
  float s = 0.f;
  for (int x = threadIdx.x & 63; x < W; x += 64) s += row[x];
  return wave_sum(s) / (float)W;
}

// Window of the padded plane held in LDS: padded rows [wy0, wy0+WH), padded cols [wx0, wx0+WW).
struct Window {
  int wy0, wx0, WH, WW;
};

// stage src plane (image H x W) into the window through the geocyclic map; subst: replace source
// rows 0 / H-1 by the given means (tiled schedule; WHOLE computes the means in LDS afterwards)
__device__ __forceinline__ void stage_window(float* win, const float* __restrict__ F, const Window& w,
                                             int H, int W, int p, bool subst, float m0, float m1) {
  const int Hp = H + 2 * p;
  for (int lr = threadIdx.x >> 6; lr < w.WH; lr += 4) {
    const int r = w.wy0 + lr;  // padded row
    float* dst = win + lr * w.WW;
    if (r < 0 || r >= Hp) {
      for (int lc = threadIdx.x & 63; lc < w.WW; lc += 64) dst[lc] = 0.f;
      continue;
    }
    const int ii = r - p;
    int sr;
    bool mir = false;
    if (ii < 0) { sr = -ii; mir = true; }
    else if (ii >= H) { sr = 2 * (H - 1) - ii; mir = true; }
    else sr = ii;
    const float* srow = F + (int64_t)sr * W;
    const bool pole0 = subst && sr == 0, pole1 = subst && sr == H - 1;
    // periodic in longitude for any window offset: one modulo per row, then incremental wraps
    int jj = (w.wx0 + (int)(threadIdx.x & 63) - p + (mir ? (W >> 1) : 0)) % W;
    if (jj < 0) jj += W;
    const int step = 64 % W;
    for (int lc = threadIdx.x & 63; lc < w.WW; lc += 64, jj += step) {
      if (jj >= W) jj -= W;
      float val = srow[jj];
      if (pole0) val = m0;
      if (pole1) val = m1;
      dst[lc] = val;
    }
  }
}

// iterate i = tid, tid+256, ... < th*tw as (yl, xl) without a division per point
struct TileIter {
  int yl, xl, dy, dx, tw;
  __device__ __forceinline__ TileIter(int tid, int tw_) : tw(tw_) {
    yl = tid / tw_; xl = tid - yl * tw_; dy = 256 / tw_; dx = 256 - dy * tw_;
  }
  __device__ __forceinline__ void next() {
    yl += dy; xl += dx;
    if (xl >= tw) { xl -= tw; ++yl; }
  }
};

// ======================================================================================
// forward
// ======================================================================================
template <int MODE, bool WHOLE>
__global__ void __launch_bounds__(256)
sl_advect_fwd_kernel(const float* __restrict__ field, const float* __restrict__ u,
                     const float* __restrict__ v, float* __restrict__ out,
                     const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lon, const float* __restrict__ fmeans, int K,
                     AdvGeom g, int64_t f_bs, int64_t uv_bs, int64_t o_bs, int halo, int tiles_x,
                     int tiles, int vec4) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = Interp<MODE>::NT;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int plane = WHOLE ? blockIdx.x : blockIdx.x / tiles;
  const int tile = WHOLE ? 0 : blockIdx.x - plane * tiles;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;

  const int ty0 = WHOLE ? 0 : (tile / tiles_x) * TILE_H, tx0 = WHOLE ? 0 : (tile % tiles_x) * TILE_W;
  const int th = WHOLE ? H : min(TILE_H, H - ty0), tw = WHOLE ? W : min(TILE_W, W - tx0);
  Window w;
  if (WHOLE) { w.wy0 = 0; w.wx0 = 0; w.WH = Hp; w.WW = Wp; }
  else { w.wy0 = ty0 + p - halo; w.wx0 = tx0 + p - halo; w.WH = TILE_H + 2 * halo + NT; w.WW = TILE_W + 2 * halo + NT; }
  float* win = smem;                       // [WH*WW]
  float* pole_out = smem + w.WH * w.WW;    // [2*W]  (WHOLE only)

  float m0 = 0.f, m1 = 0.f;
  if (!WHOLE) { m0 = fmeans[2 * plane]; m1 = fmeans[2 * plane + 1]; }
  stage_window(win, F, w, H, W, p, !WHOLE, m0, m1);
  __syncthreads();
  if (WHOLE) {
    if (wave < 2) {  // pole rows <- their mean, over the whole padded row (lon halo included)
      float* row = win + (wave == 0 ? p : H - 1 + p) * Wp;
      const float m = wave_row_mean(row + p, W);
      for (int x = tid & 63; x < Wp; x += 64) row[x] = m;
    }
    __syncthreads();
  }

  const int npts = th * tw;
  // (a 4-points-per-thread variant with 16-B loads/stores measured SLOWER: 124 VGPRs halve the
  //  occupancy and the kernel is latency/issue-bound, not bandwidth-bound)
  (void)vec4;
  // one arrival point: departure -> tap block -> window gather (or L2 fallback in the tiled schedule)
  auto point = [&](float uu, float vv, float sa, float ca, float lo) -> float {
    float ix, iy, tx, ty, wx[NT], wy[NT];
    int bx, by, sx, sy;
    departure(uu, vv, sa, ca, lo, g, ix, iy, nullptr);
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    shift_weights<NT>(wx, sx);
    shift_weights<NT>(wy, sy);
    int ry = by - w.wy0, rx = bx - w.wx0;
    bool inwin = true;
    if (!WHOLE) {
      if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
      inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    }
    float acc = 0.f;
    if (inwin) {
      const float* base = win + ry * w.WW + rx;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) rowacc += base[a * w.WW + bb] * wx[bb];
        acc += rowacc * wy[a];
      }
    } else {  // tiled schedule only: taps served by L2 through the index map
      const int lastrow = H - 1;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
          int r, c;
          geo_src(by + a - p, bx + bb - p, H, W, r, c);
          float val = F[(int64_t)r * W + c];
          if (r == 0) val = m0; else if (r == lastrow) val = m1;
          rowacc += val * wx[bb];
        }
        acc += rowacc * wy[a];
      }
    }
    return acc;
  };
  {
    TileIter it(tid, tw);
    for (int i = tid; i < npts; i += 256, it.next()) {
      const int y = ty0 + it.yl, x = tx0 + it.xl, idx = y * W + x;
      const float acc = point(U[idx], V[idx], sin_lat[idx], cos_lat[idx], lon[idx]);
      if (WHOLE && y == 0) pole_out[x] = acc;
      else if (WHOLE && y == H - 1) pole_out[W + x] = acc;
      else O[idx] = acc;
    }
  }
  if (WHOLE) {
    __syncthreads();
    if (wave < 2) {
      const float* row = pole_out + (wave == 0 ? 0 : W);
      const float m = wave_row_mean(row, W);
      float* orow = O + (wave == 0 ? 0 : (int64_t)(H - 1) * W);
      for (int x = tid & 63; x < W; x += 64) orow[x] = m;
    }
  }
}

// ======================================================================================
// backward
// ======================================================================================
__device__ __forceinline__ void fixed_point_scale(float mx, float& scale, float& inv) {
  scale = 0.f; inv = 0.f;
  if (mx > 0.f && mx < INFINITY) {
    int e = 0;
    frexpf(mx, &e);                       // mx < 2^e
    e = e < -80 ? -80 : (e > 80 ? 80 : e);
    scale = ldexpf(1.0f, 40 - e);
    inv = ldexpf(1.0f, e - 40);
  } else if (!(mx < INFINITY)) {
    inv = NAN;                            // non-finite cotangent: propagate NaN like float adds would
  }
}

__device__ __forceinline__ void departure_backward(const DepState& st, float sa, float ca, float gix,
                                                   float giy, float kx, float ky, float dt, float& gu,
                                                   float& gv) {
  const float glam_c = gix * kx, gphi_c = giy * ky;
  const float sc = fminf(fmaxf(st.s, -CLAMP_HI), CLAMP_HI);
  const float gs = (st.s >= -CLAMP_HI && st.s <= CLAMP_HI) ? gphi_c / sqrtf(1.0f - sc * sc) : 0.f;
  const float den = st.n * st.n + st.d * st.d;
  const float gn = glam_c * st.d / den;
  const float gd = -glam_c * st.n / den;
  const float gphi = gs * (st.cp * ca - st.sp * st.cl * sa) + gn * (-st.sp * st.sl) +
                     gd * (-st.sp * st.cl * ca - st.cp * sa);
  const float glam = gs * (-st.cp * st.sl * sa) + gn * (st.cp * st.cl) + gd * (-st.cp * st.sl * ca);
  gu = -dt * glam;
  gv = -dt * gphi;
}

template <int MODE, bool WHOLE>
__global__ void __launch_bounds__(256)
sl_advect_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ field,
                     const float* __restrict__ u, const float* __restrict__ v,
                     float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                     const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lon, const float* __restrict__ fmeans,
                     const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                     int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, int halo, int tiles_x, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = Interp<MODE>::NT;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int plane = WHOLE ? blockIdx.x : blockIdx.x / tiles;
  const int tile = WHOLE ? 0 : blockIdx.x - plane * tiles;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;

  const int ty0 = WHOLE ? 0 : (tile / tiles_x) * TILE_H, tx0 = WHOLE ? 0 : (tile % tiles_x) * TILE_W;
  const int th = WHOLE ? H : min(TILE_H, H - ty0), tw = WHOLE ? W : min(TILE_W, W - tx0);
  Window w;
  if (WHOLE) { w.wy0 = 0; w.wx0 = 0; w.WH = Hp; w.WW = Wp; }
  else { w.wy0 = ty0 + p - halo; w.wx0 = tx0 + p - halo; w.WH = TILE_H + 2 * halo + NT; w.WW = TILE_W + 2 * halo + NT; }
  const int wn = w.WH * w.WW, wn2 = (wn + 1) & ~1;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);  // [wn] fixed-point sums
  float* win = smem + 2 * wn2;                                             // [wn]  F~ window
  float* misc = win + wn2;   // [0..1] pole means of gout, [2..5] per-wave max, [6] scale, [7] 1/scale

  float m0 = 0.f, m1 = 0.f, gm0 = 0.f, gm1 = 0.f;
  if (!WHOLE) {
    m0 = fmeans[2 * plane]; m1 = fmeans[2 * plane + 1];
    gm0 = gmeans[2 * plane]; gm1 = gmeans[2 * plane + 1];
  }
  stage_window(win, F, w, H, W, p, !WHOLE, m0, m1);
  for (int i = tid; i < wn; i += 256) acc[i] = 0ull;
  // max |cotangent| over this workgroup's arrival points -> fixed-point scale
  const int npts = th * tw;
  float gmax = WHOLE ? 0.f : fmaxf(fabsf(gm0), fabsf(gm1));
  for (int i = tid; i < npts; i += 256) {
    const int yl = i / tw, xl = i - yl * tw;
    gmax = fmaxf(gmax, fabsf(GO[(ty0 + yl) * W + tx0 + xl]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
  if ((tid & 63) == 0) misc[2 + wave] = gmax;
  __syncthreads();
  if (WHOLE) {
    if (wave < 2) {
      float* row = win + (wave == 0 ? p : H - 1 + p) * Wp;
      const float m = wave_row_mean(row + p, W);
      for (int x = tid & 63; x < Wp; x += 64) row[x] = m;
    } else {
      // adjoint of the final pole mean: the cotangent of a pole row is its own row mean
      const float* row = GO + (wave == 2 ? 0 : (int64_t)(H - 1) * W);
      const float m = wave_row_mean(row, W);
      if ((tid & 63) == 0) misc[wave - 2] = m;
    }
  }
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(misc[2], misc[3]), fmaxf(misc[4], misc[5]));
    fixed_point_scale(mx, misc[6], misc[7]);
  }
  __syncthreads();
  if (WHOLE) { gm0 = misc[0]; gm1 = misc[1]; }
  const float scale = misc[6];

  const float kx = ((float)W - 1.0f) / g.d_lon, ky = ((float)H - 1.0f) / g.d_lat;
  TileIter it(tid, tw);
  for (int i = tid; i < npts; i += 256, it.next()) {
    const int y = ty0 + it.yl, x = tx0 + it.xl, idx = y * W + x;
    const float sa = sin_lat[idx], ca = cos_lat[idx];
    float ix, iy, tx, ty, wx[NT], wy[NT], dwx[NT], dwy[NT];
    int bx, by, sx, sy;
    DepState st;
    departure(U[idx], V[idx], sa, ca, lon[idx], g, ix, iy, &st);
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    Interp<MODE>::dweights(tx, dwx);
    Interp<MODE>::dweights(ty, dwy);
    shift_weights<NT>(wx, sx);
    shift_weights<NT>(dwx, sx);
    shift_weights<NT>(wy, sy);
    shift_weights<NT>(dwy, sy);
    const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : GO[idx]);
    const float gs_ = gval * scale;
    int ry = by - w.wy0, rx = bx - w.wx0;
    bool inwin = true;
    if (!WHOLE) {
      if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
      inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    }
    float gix = 0.f, giy = 0.f;
    if (inwin) {
      const int base = ry * w.WW + rx;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float sxv = 0.f, sdx = 0.f;
        const float gwy = gs_ * wy[a];
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
          const int cell = base + a * w.WW + bb;
          const float val = win[cell];
#ifndef ADV_NO_ATOMIC
          atomicAdd(&acc[cell], (unsigned long long)__float2ll_rn(gwy * wx[bb]));
#endif
          sxv += val * wx[bb];
          sdx += val * dwx[bb];
        }
        gix += wy[a] * sdx;
        giy += dwy[a] * sxv;
      }
    } else {  // tiled schedule only
      const int lastrow = H - 1;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float sxv = 0.f, sdx = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) {
          int r, c;
          geo_src(by + a - p, bx + bb - p, H, W, r, c);
          float val = F[(int64_t)r * W + c];
          if (r == 0) val = m0; else if (r == lastrow) val = m1;
          atomicAdd(&GF[(int64_t)r * W + c], gval * wy[a] * wx[bb]);
          sxv += val * wx[bb];
          sdx += val * dwx[bb];
        }
        gix += wy[a] * sdx;
        giy += dwy[a] * sxv;
      }
    }
    gix *= gval;
    giy *= gval;
    float guv, gvv;
    departure_backward(st, sa, ca, gix, giy, kx, ky, g.dt, guv, gvv);
    GU[idx] = guv;
    GV[idx] = gvv;
  }
  __syncthreads();
  const double inv = (double)misc[7];
  if (WHOLE) {
    // fold the halo back: every source cell sums its aliases (adjoint of the a1 map), then the
    // adjoint of the first pole mean; the float plane reuses the window storage
    for (int i = tid; i < P; i += 256) {
      const int y = i / W, x = i - y * W;
      long long s = 0;
      geo_for_each_alias(y, x, H, W, p, [&](int ii, int jj) { s += (long long)acc[(ii + p) * Wp + jj + p]; });
      win[i] = (float)((double)s * inv);
    }
    __syncthreads();
    if (wave < 2) {
      float* row = win + (wave == 0 ? 0 : (H - 1) * W);
      const float m = wave_row_mean(row, W);
      for (int x = tid & 63; x < W; x += 64) row[x] = m;
    }
    __syncthreads();
    for (int i = tid; i < P; i += 256) GF[i] = win[i];
  } else {
    // flush the window once: one global float atomic per touched cell instead of 16 per point
    for (int i = tid; i < wn; i += 256) {
      const long long s = (long long)acc[i];
      if (s == 0) continue;
      const int lr = i / w.WW, lc = i - lr * w.WW;
      const int r = w.wy0 + lr;
      if (r < 0 || r >= Hp) continue;
      int jj = (w.wx0 + lc - p) % W;
      if (jj < 0) jj += W;
      int sr, sc;
      geo_src(r - p, jj, H, W, sr, sc);
      atomicAdd(&GF[(int64_t)sr * W + sc], (float)((double)s * inv));
    }
  }
}

// ---- pole-row helpers of the tiled schedule ---------------------------------------------
__global__ void __launch_bounds__(256)
pole_row_means(const float* __restrict__ src, float* __restrict__ means, int planes, int K, int H,
               int W, int64_t bs) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;  // one wave per (plane,row)
  if (w >= planes * 2) return;
  const int plane = w >> 1, which = w & 1;
  const int b = plane / K, k = plane - b * K;
  const float* row = src + (int64_t)b * bs + (int64_t)k * H * W + (which ? (int64_t)(H - 1) * W : 0);
  const float m = wave_row_mean(row, W);
  if ((threadIdx.x & 63) == 0) means[w] = m;
}

__global__ void __launch_bounds__(256)
pole_rows_to_mean(float* __restrict__ dst, int planes, int K, int H, int W, int64_t bs) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (w >= planes * 2) return;
  const int plane = w >> 1, which = w & 1;
  const int b = plane / K, k = plane - b * K;
  float* row = dst + (int64_t)b * bs + (int64_t)k * H * W + (which ? (int64_t)(H - 1) * W : 0);
  const float m = wave_row_mean(row, W);
  for (int x = threadIdx.x & 63; x < W; x += 64) row[x] = m;
}

int check_adv(const char* name, int B, int K, int H, int W, int mode) {
  PD_REQUIRE(B >= 0 && K >= 1 && H >= 4 && W >= 4, "%s: bad shape B=%d K=%d H=%d W=%d", name, B, K, H, W);
  PD_REQUIRE(W % 2 == 0, "%s: Number of longitude points must be even", name);
  PD_REQUIRE(mode == PARADIS_INTERP_BILINEAR || mode == PARADIS_INTERP_BICUBIC,
             "%s: interpolation mode must be 1 (bilinear) or 2 (bicubic)", name);
  PD_REQUIRE((int64_t)B * K < (1 << 30) && (int64_t)H * W < (1ll << 30), "%s: too large", name);
  return 0;
}

constexpr size_t WHOLE_LDS_LIMIT = 64 * 1024;
int g_force_tiled = -1;   // debug: -1 auto, 0/1 force
int g_halo = 6;           // window halo (padded cells) of the tiled schedule

bool use_tiled(size_t whole_bytes) {
  if (g_force_tiled == 1) return true;
  return whole_bytes > WHOLE_LDS_LIMIT;
}

}  // namespace

// test/diagnostic hooks: force the tiled schedule regardless of plane size; set its halo
extern "C" void paradis_debug_set_advect_gmem(int on) { g_force_tiled = on; }
extern "C" void paradis_debug_set_advect_halo(int halo) { g_halo = halo < 0 ? 0 : (halo > 16 ? 16 : halo); }

extern "C" size_t paradis_sl_advect_ws_bytes(int B, int K, int H, int W) {
  (void)H; (void)W;
  return (size_t)B * K * 4 * sizeof(float) + 256;
}

#define ADV_LAUNCH(KERNEL, WHOLE_, grid, lds, ...)                                                  \
  do {                                                                                              \
    if (mode == PARADIS_INTERP_BICUBIC)                                                             \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BICUBIC, WHOLE_>), dim3(grid), dim3(256), lds, st,  \
                         __VA_ARGS__);                                                              \
    else                                                                                            \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BILINEAR, WHOLE_>), dim3(grid), dim3(256), lds, st, \
                         __VA_ARGS__);                                                              \
  } while (0)

extern "C" int paradis_sl_advect_fwd(const float* field, const float* u, const float* v, float* out,
                                     const float* sin_lat, const float* cos_lat, const float* lon,
                                     int B, int K, int H, int W, int64_t f_bs, int64_t uv_bs,
                                     int64_t o_bs, float dt, float min_lat, float min_lon,
                                     float d_lat, float d_lon, int mode, void* workspace,
                                     void* stream) {
  if (int e = check_adv("sl_advect_fwd", B, K, H, W, mode)) return e;
  if (B == 0) return 0;
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1, NT = 2 * p;
  AdvGeom g{H, W, p, dt, min_lat, min_lon, d_lat, d_lon};
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K;
  const size_t whole = ((size_t)(H + 2 * p) * (W + 2 * p) + 2 * W) * sizeof(float);
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const int vec4 = (W % 4 == 0) && (uv_bs % 4 == 0) && (o_bs % 4 == 0) && a16(u) && a16(v) && a16(out) &&
                   a16(sin_lat) && a16(cos_lat) && a16(lon);
  if (!use_tiled(whole)) {
    ADV_LAUNCH(sl_advect_fwd_kernel, true, planes, whole, field, u, v, out, sin_lat, cos_lat, lon,
               (const float*)nullptr, K, g, f_bs, uv_bs, o_bs, 0, 1, 1, vec4);
    PD_CHECK_LAUNCH("sl_advect_fwd");
    return 0;
  }
  PD_REQUIRE(workspace != nullptr, "sl_advect_fwd: workspace required for the tiled schedule");
  float* fmeans = (float*)workspace;
  const int mean_blocks = (planes * 2 * 64 + 255) / 256;
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, field, fmeans, planes, K, H, W, f_bs);
  const int tx = (W + TILE_W - 1) / TILE_W, ty = (H + TILE_H - 1) / TILE_H, tiles = tx * ty;
  PD_REQUIRE((int64_t)planes * tiles < (1ll << 31), "sl_advect_fwd: too many tiles");
  const size_t lds = (size_t)(TILE_H + 2 * g_halo + NT) * (TILE_W + 2 * g_halo + NT) * sizeof(float);
  ADV_LAUNCH(sl_advect_fwd_kernel, false, (unsigned)(planes * tiles), lds, field, u, v, out, sin_lat,
             cos_lat, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, g_halo, tx, tiles, vec4);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, out, planes, K, H, W, o_bs);
  PD_CHECK_LAUNCH("sl_advect_fwd(tiled)");
  return 0;
}

extern "C" int paradis_sl_advect_bwd(const float* gout, const float* field, const float* u,
                                     const float* v, float* gfield, float* gu, float* gv,
                                     const float* sin_lat, const float* cos_lat, const float* lon,
                                     int B, int K, int H, int W, int64_t go_bs, int64_t f_bs,
                                     int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, float dt,
                                     float min_lat, float min_lon, float d_lat, float d_lon, int mode,
                                     void* workspace, void* stream) {
  if (int e = check_adv("sl_advect_bwd", B, K, H, W, mode)) return e;
  if (B == 0) return 0;
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1, NT = 2 * p;
  AdvGeom g{H, W, p, dt, min_lat, min_lon, d_lat, d_lon};
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K, P = H * W;
  auto lds_of = [](size_t cells) { return (3 * ((cells + 1) & ~(size_t)1) + 8) * sizeof(float); };
  const size_t whole = lds_of((size_t)(H + 2 * p) * (W + 2 * p));
  if (!use_tiled(whole)) {
    ADV_LAUNCH(sl_advect_bwd_kernel, true, planes, whole, gout, field, u, v, gfield, gu, gv, sin_lat,
               cos_lat, lon, (const float*)nullptr, (const float*)nullptr, K, g, go_bs, f_bs, uv_bs,
               gf_bs, guv_bs, 0, 1, 1);
    PD_CHECK_LAUNCH("sl_advect_bwd");
    return 0;
  }
  PD_REQUIRE(workspace != nullptr, "sl_advect_bwd: workspace required for the tiled schedule");
  PD_REQUIRE(gf_bs == (int64_t)K * P, "sl_advect_bwd: tiled schedule needs a contiguous gfield");
  float* fmeans = (float*)workspace;
  float* gmeans = fmeans + (size_t)planes * 2;
  const int mean_blocks = (planes * 2 * 64 + 255) / 256;
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, field, fmeans, planes, K, H, W, f_bs);
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, gout, gmeans, planes, K, H, W, go_bs);
  if (hipMemsetAsync(gfield, 0, (size_t)planes * P * sizeof(float), st) != hipSuccess) {
    paradis_set_error("sl_advect_bwd: memset failed");
    return 2;
  }
  const int tx = (W + TILE_W - 1) / TILE_W, ty = (H + TILE_H - 1) / TILE_H, tiles = tx * ty;
  PD_REQUIRE((int64_t)planes * tiles < (1ll << 31), "sl_advect_bwd: too many tiles");
  const size_t lds = lds_of((size_t)(TILE_H + 2 * g_halo + NT) * (TILE_W + 2 * g_halo + NT));
  ADV_LAUNCH(sl_advect_bwd_kernel, false, (unsigned)(planes * tiles), lds, gout, field, u, v, gfield,
             gu, gv, sin_lat, cos_lat, lon, (const float*)fmeans, (const float*)gmeans, K, g, go_bs,
             f_bs, uv_bs, gf_bs, guv_bs, g_halo, tx, tiles);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, gfield, planes, K, H, W, gf_bs);
  PD_CHECK_LAUNCH("sl_advect_bwd(tiled)");
  return 0;
}
