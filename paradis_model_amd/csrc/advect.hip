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
// shipped library defines none):  ADV_NO_ATOMIC, ADV_NO_TRIG, ADV_OCML_SINCOS, ADV_NO_SMALL_ANGLE, ADV_IEEE_DIV, ADV_OCML_ATAN2, ...
namespace {

constexpr float TWO_PI_F = 6.283185307179586f;
constexpr float CLAMP_HI = 0.9999999f;  // float(1 - 1e-7), as torch.clamp converts its python bound
constexpr float KA = -0.75f;
#ifndef ADV_TILE_HB
#define ADV_TILE_HB 16
#endif
#ifndef ADV_THREADS_B
#define ADV_THREADS_B 512
#endif
constexpr int TILE_H = ADV_TILE_HB, TILE_W = 128;  // arrival tile of the tiled schedule (backward)
#ifndef ADV_TILE_HF
#define ADV_TILE_HF 64
#endif
// forward tile height: a taller tile amortises the halo (window cells per arrival point 2.6 at 16
// rows, 1.9 at 32, 1.5 at 64 with a halo of 8); the forward window is 4 B/cell, so LDS is not the
// limit.  Measured at 128x256: 1.64 / 1.42 / 1.23 ms per launch for 16 / 32 / 64 rows
constexpr int TILE_HF = ADV_TILE_HF;
// threads per tile in the tiled schedule: the window fixes the LDS per workgroup, so waves per SIMD
// come from the workgroup size.  Backward (12 B/cell, 2 workgroups per CU): at 256 threads it ran 1.7
// waves per SIMD at 29 % VALU issue, 512 threads measured 6.9 -> 5.6 ms at 128x256; the forward
// (4 B/cell) has the occupancy already and is 5-10 % faster with 256.
constexpr int TILED_THREADS_FWD = TILE_HF >= 32 ? 512 : 256, TILED_THREADS_BWD = ADV_THREADS_B;
#ifndef ADV_UNROLL
#define ADV_UNROLL 4
#endif
#ifndef ADV_UNROLL_BWD
#define ADV_UNROLL_BWD 2
#endif
#ifndef ADV_PF
#define ADV_PF 2   // prefetch distance (points) of the operand loads in the whole-plane forward kernel
#endif

struct AdvGeom {
  int H, W, p;
  float dt, min_lat, min_lon, d_lat, d_lon;
  // correctly rounded reciprocals of the four loop-invariant divisors (host, via double)
  float r_lat, r_lon, r_wpm1, r_hpm1;
};

// x / d for a loop-invariant d with rd = RN(1/d): q = RN(x rd), r = x - q d (exact in the FMA),
// q' = RN(q + r rd) is the correctly rounded quotient (Markstein), i.e. bit-identical to the IEEE
// division of the reference's fp32 chain at 3 instructions instead of ~10.  Verified bit-exact
// against '/' for the divisors of every grid (tests/test_hip_pad_advect.py, oracle/check_div.c).
__device__ __forceinline__ float div_by(float x, float d, float rd) {
#ifdef ADV_IEEE_DIV
  (void)rd;
  return x / d;
#else
  const float q = x * rd;
  const float r = fmaf(-q, d, x);
  return fmaf(r, rd, q);
#endif
}

struct DepState {  // intermediates needed by the backward chain
  float sp, cp, sl, cl, s, n, d;
};

// sin and cos: cephes minimax polynomials on [-pi/4, pi/4] (<= ~1 ulp) behind a Cody-Waite reduction
// (fdlibm's float split of pi/2); huge arguments take the ocml path.
__device__ __forceinline__ void sincos_kernel(float r, float& ps, float& pc) {
  const float z = r * r;
  ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
            fmaf(-0.5f, z, 1.0f));
}

__device__ __forceinline__ void sincos_fast(float x, float& s, float& c) {
  if (fabsf(x) > 8192.0f) {
    sincosf(x, &s, &c);
    return;
  }
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5707855225e+00f, x);   // pi/2 split in three parts with trailing zero bits
  r = fmaf(-k, 1.0804273188e-05f, r);
  r = fmaf(-k, 6.0770999344e-11f, r);
  const int q = (int)k;
  float ps, pc;
  sincos_kernel(r, ps, pc);
  const float ss = (q & 1) ? pc : ps;
  const float cc = (q & 1) ? ps : pc;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// sin/cos of the two rotation angles of a point.  |angle| < 0.78 (< pi/4: the reduction's k is 0 and
// r = x exactly) for every lane of the wave is the normal case - displacements of less than 45 degrees
// per step - and needs no reduction and no quadrant selects: same bits, 9 instead of ~35
// instructions per angle.
__device__ __forceinline__ void sincos_pair(float phi, float lam, float& sp, float& cp, float& sl, float& cl) {
#if defined(ADV_OCML_SINCOS)
  sincosf(phi, &sp, &cp);
  sincosf(lam, &sl, &cl);
#else
#ifndef ADV_NO_SMALL_ANGLE
  if (__all(fabsf(phi) < 0.78f && fabsf(lam) < 0.78f)) {
    sincos_kernel(phi, sp, cp);
    sincos_kernel(lam, sl, cl);
    return;
  }
#endif
  sincos_fast(phi, sp, cp);
  sincos_fast(lam, sl, cl);
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

// atan2f for finite arguments of ordinary magnitude (here n^2 + d^2 = cos^2(lat_d) > 1e-7): the
// ocml algorithm (min/max quotient by v_rcp, degree-8 minimax in t^2, octant fix-ups) without its
// frexp/ldexp overflow scaling and inf/NaN classification - same bits in this range.
__device__ __forceinline__ float atan2_finite(float y, float x) {
#ifdef ADV_OCML_ATAN2
  return atan2f(y, x);
#else
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float z = t * t;
  float pp = fmaf(z, 0x1.5a54bp-9f, -0x1.f4b218p-7f);
  pp = fmaf(z, pp, 0x1.53f67ep-5f);
  pp = fmaf(z, pp, -0x1.2fa9aep-4f);
  pp = fmaf(z, pp, 0x1.b26364p-4f);
  pp = fmaf(z, pp, -0x1.22c1ccp-3f);
  pp = fmaf(z, pp, 0x1.99717ep-3f);
  pp = fmaf(z, pp, -0x1.5554c4p-2f);
  float a = fmaf(t, z * pp, t);
  a = (ay > ax) ? 0x1.921fb6p+0f - a : a;
  a = (x < 0.f) ? 0x1.921fb6p+1f - a : a;
  a = (y == 0.f) ? ((__float_as_int(x) < 0) ? 0x1.921fb6p+1f : 0.f) : a;   // also covers 0/0
  return copysignf(a, y);
#endif
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
  sincos_pair(phi, lam, sp, cp, sl, cl);
  const float cc = cp * cl;
  const float s = sp * ca + cc * sa;
  const float sc = fminf(fmaxf(s, -CLAMP_HI), CLAMP_HI);
  const float lat_d = asinf(sc);
  const float n = cp * sl;
  const float d = cc * ca - sp * sa;
  float lon_d = lon_a + atan2_finite(n, d);
  lon_d = lon_d + TWO_PI_F;
  const float m = wrap_two_pi(lon_d);
  const float pix_x = div_by(m - g.min_lon, g.d_lon, g.r_lon) * ((float)g.W - 1.0f);
  const float pix_y = div_by(lat_d - g.min_lat, g.d_lat, g.r_lat) * ((float)g.H - 1.0f);
  const float wpm1 = (float)(g.W + 2 * g.p - 1), hpm1 = (float)(g.H + 2 * g.p - 1);
  const float gx = 2.0f * div_by(pix_x + (float)g.p, wpm1, g.r_wpm1) - 1.0f;
  const float gy = 2.0f * div_by(pix_y + (float)g.p, hpm1, g.r_hpm1) - 1.0f;
  ix = ((gx + 1.0f) / 2.0f) * wpm1;
  iy = ((gy + 1.0f) / 2.0f) * hpm1;
  if (st) {
    st->sp = sp; st->cp = cp; st->sl = sl; st->cl = cl; st->s = s; st->n = n; st->d = d;
  }
}

// (explicit FMAs: the weights and tap sums are not coordinate-critical - an ulp of a weight is
//  1e-7 relative in the result, whereas an ulp of a sample coordinate is multiplied by the field slope)
__device__ __forceinline__ float cub1(float x) { return fmaf(fmaf(KA + 2.f, x, -(KA + 3.f)) * x, x, 1.f); }
__device__ __forceinline__ float cub2(float x) { return fmaf(fmaf(fmaf(KA, x, -5.f * KA), x, 8.f * KA), x, -4.f * KA); }
__device__ __forceinline__ float dcub1(float x) { return fmaf(3.f * (KA + 2.f), x, -2.f * (KA + 3.f)) * x; }
__device__ __forceinline__ float dcub2(float x) { return fmaf(fmaf(3.f * KA, x, -10.f * KA), x, 8.f * KA); }

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
  // NaN/inf/huge: v_med3 clamps (NaN -> the lower bound), the shift then zeroes every weight
  const int x0 = (int)__builtin_amdgcn_fmed3f(x0f, -8.0f, (float)(Wp + 8)) + OFF0;
  const int y0 = (int)__builtin_amdgcn_fmed3f(y0f, -8.0f, (float)(Hp + 8)) + OFF0;
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

// iterate i = tid, tid+256, ... < th*tw as (yl, xl) without a division per point
struct TileIter {
  int yl, xl, dy, dx, tw;
  __device__ __forceinline__ TileIter(int tid, int tw_, int nth = 256) : tw(tw_) {
    yl = tid / tw_; xl = tid - yl * tw_; dy = nth / tw_; dx = nth - dy * tw_;
  }
  __device__ __forceinline__ void next() {
    yl += dy; xl += dx;
    if (xl >= tw) { xl -= tw; ++yl; }
  }
};

// stage src plane (image H x W) into the window through the geocyclic map; subst: replace source
// rows 0 / H-1 by the given means (tiled schedule; WHOLE computes the means in LDS afterwards).
// Flat over the window in batches: all loads of a batch are issued before the first LDS write, so a
// workgroup pays ~one memory round trip for its window.  (A row-per-wave loop serialised one round
// trip per row - 9 per wave at 32x64 - and cost 23 % of the forward kernel.)
constexpr int STAGE_BATCH = 6;
__device__ __forceinline__ void stage_window(float* win, const float* __restrict__ F, const Window& w,
                                             int H, int W, int p, bool subst, float m0, float m1,
                                             int nth = 256) {
  // A thread keeps one window column (its longitude wrap - plain and mirrored - is computed once) and
  // walks down the rows; only the cheap row map (mirror beyond a pole) is per element.  A flat
  // element-per-thread assignment paid the full index map (~50 VALU) for each of the 2.6x more window
  // cells than arrival points: 1 of the 2.1 ms of the tiled forward at 128x256.
  const int Hp = H + 2 * p;
  const int tid = threadIdx.x;
  const int cols = w.WW < nth ? w.WW : nth;          // window columns per pass
  const int rpp = nth / cols;                         // window rows per pass
  const int r0 = tid / cols, c0 = tid - r0 * cols;
  if (r0 >= rpp) return;
  for (int lc = c0; lc < w.WW; lc += cols) {
    int jj = (w.wx0 + lc - p) % W;
    if (jj < 0) jj += W;
    int jm = jj + (W >> 1);
    if (jm >= W) jm -= W;
    for (int l0 = r0; l0 < w.WH; l0 += rpp * STAGE_BATCH) {
      float val[STAGE_BATCH];
#pragma unroll
      for (int j = 0; j < STAGE_BATCH; ++j) {
        // unconditional load from a clamped (always valid) source cell, then select
        const int lr = l0 + rpp * j;
        const int r = w.wy0 + lr;                      // padded row
        const bool valid = lr < w.WH && r >= 0 && r < Hp;
        const int ii = min(max(r, 0), Hp - 1) - p;
        int sr = ii;
        bool mir = false;
        if (ii < 0) { sr = -ii; mir = true; }
        else if (ii >= H) { sr = 2 * (H - 1) - ii; mir = true; }
        float v = F[(int64_t)sr * W + (mir ? jm : jj)];
        if (subst && sr == 0) v = m0;
        if (subst && sr == H - 1) v = m1;
        val[j] = valid ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < STAGE_BATCH; ++j) {
        const int lr = l0 + rpp * j;
        if (lr < w.WH) win[lr * w.WW + lc] = val[j];
      }
    }
  }
}

// ======================================================================================
// forward
// ======================================================================================
template <int MODE, bool WHOLE, int NTH>
__global__ void __launch_bounds__(NTH)
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

  const int ty0 = WHOLE ? 0 : (tile / tiles_x) * TILE_HF, tx0 = WHOLE ? 0 : (tile % tiles_x) * TILE_W;
  const int th = WHOLE ? H : min(TILE_HF, H - ty0), tw = WHOLE ? W : min(TILE_W, W - tx0);
  Window w;
  if (WHOLE) { w.wy0 = 0; w.wx0 = 0; w.WH = Hp; w.WW = Wp; }
  else { w.wy0 = ty0 + p - halo; w.wx0 = tx0 + p - halo; w.WH = TILE_HF + 2 * halo + NT; w.WW = TILE_W + 2 * halo + NT; }
  float* win = smem;                       // [WH*WW]
  float* pole_out = smem + w.WH * w.WW;    // [2*W]  (WHOLE only)

  float m0 = 0.f, m1 = 0.f;
  if (!WHOLE) { m0 = fmeans[2 * plane]; m1 = fmeans[2 * plane + 1]; }
#ifndef ADV_NO_STAGE
  if (WHOLE && vec4) stage_plane_vec4(win, F, H, W, p);
  else stage_window(win, F, w, H, W, p, !WHOLE, m0, m1, NTH);
#endif
  __syncthreads();
  if (WHOLE) {
    // pole rows <- their mean, over the whole padded row (lon halo included).  (Reducing the two rows
    // from global before staging would save this barrier but puts a dependent load in front of the
    // staging loads: measured 2-7 % slower.)
    if (wave < 2) {
      float* row = win + (wave == 0 ? p : H - 1 + p) * Wp;
      const float m = wave_row_mean(row + p, W);
      for (int x = tid & 63; x < Wp; x += 64) row[x] = m;
    }
    __syncthreads();
  }
  // W == 64: an output row is one wave's lanes in one iteration -> its mean is a wave reduction
  const bool rowwave = WHOLE && W == 64;

  const int npts = th * tw;
  // (a 4-points-per-thread variant with 16-B loads/stores measured SLOWER: 124 VGPRs halve the
  //  occupancy and the kernel is latency/issue-bound, not bandwidth-bound)
  // one arrival point: departure -> tap block -> window gather (or L2 fallback in the tiled schedule)
  auto point = [&](float uu, float vv, float sa, float ca, float lo) -> float {
    float ix, iy, tx, ty, wx[NT], wy[NT];
    int bx, by, sx, sy;
    departure(uu, vv, sa, ca, lo, g, ix, iy, nullptr);
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    if (sx | sy) {   // only when a coordinate rounds onto the plane edge (or is not finite)
      shift_weights<NT>(wx, sx);
      shift_weights<NT>(wy, sy);
    }
    int ry = by - w.wy0, rx = bx - w.wx0;
    bool inwin = true;
    if (!WHOLE) {
      if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
      inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    }
    float acc = 0.f;
    if (inwin) {
      const float* base = win + ry * w.WW + rx;
#ifdef ADV_NO_GATHER
      acc = (float)(ry * w.WW + rx) * (wx[0] + wx[1] + wy[0] + wy[1] + wx[NT - 1] + wy[NT - 1]);
      (void)base;
#else
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * w.WW + bb], wx[bb], rowacc);
        acc = fmaf(rowacc, wy[a], acc);
      }
#endif
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
  if constexpr (WHOLE) {
    // Whole plane: the arrival index is the flat index.  Operands are prefetched ADV_PF points ahead
    // into a queue whose slots are fixed registers (the loop is unrolled ADV_PF times) with
    // unconditional, clamped loads: straight-line code lets the compiler count vmcnt exactly -
    // with per-lane conditionals around the loads it emitted s_waitcnt vmcnt(0) every iteration,
    // i.e. every point waited for the previous point's store to be acknowledged.
    const int last = npts - 1;
    float qu[ADV_PF], qv[ADV_PF], qs[ADV_PF], qc[ADV_PF], ql[ADV_PF];
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) {
      const int j = min(tid + NTH * d, last);
      qu[d] = U[j]; qv[d] = V[j]; qs[d] = sin_lat[j]; qc[d] = cos_lat[j]; ql[d] = lon[j];
    }
    const int lastrow0 = (H - 1) * W;
    for (int i0 = tid; i0 < npts; i0 += NTH * ADV_PF) {
#pragma unroll
      for (int d = 0; d < ADV_PF; ++d) {
        const int i = i0 + NTH * d;
        const float cu = qu[d], cv = qv[d], csa = qs[d], cca = qc[d], clo = ql[d];
        {
          const int j = min(i + NTH * ADV_PF, last);
          qu[d] = U[j]; qv[d] = V[j];
#ifndef ADV_NO_TABLES
          qs[d] = sin_lat[j]; qc[d] = cos_lat[j]; ql[d] = lon[j];
#endif
        }
        const float acc = point(cu, cv, csa, cca, clo);
        if (i < npts) {
          if (i < W || i >= lastrow0) {
            if (rowwave) O[i] = wave_sum(acc) / (float)W;
            else pole_out[i < W ? i : W + i - lastrow0] = acc;
          } else {
            O[i] = acc;
          }
        }
      }
    }
  } else {
    // tiled schedule: operands of point i+1 are loaded before point i is computed
    TileIter it(tid, tw, NTH);
    float nu = 0.f, nv = 0.f, nsa = 0.f, nca = 0.f, nlo = 0.f;
    if (tid < npts) {
      const int idx = (ty0 + it.yl) * W + tx0 + it.xl;
      nu = U[idx]; nv = V[idx]; nsa = sin_lat[idx]; nca = cos_lat[idx]; nlo = lon[idx];
    }
    for (int i = tid; i < npts; i += NTH) {
      const int idx = (ty0 + it.yl) * W + tx0 + it.xl;
      const float cu = nu, cv = nv, csa = nsa, cca = nca, clo = nlo;
      it.next();
      if (i + NTH < npts) {
        const int nidx = (ty0 + it.yl) * W + tx0 + it.xl;
        nu = U[nidx]; nv = V[nidx]; nsa = sin_lat[nidx]; nca = cos_lat[nidx]; nlo = lon[nidx];
      }
      O[idx] = point(cu, cv, csa, cca, clo);
    }
  }
  if (WHOLE && !rowwave) {
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
// round-to-nearest-even integer of x (|x| < 2^51) as a two's-complement 64-bit pattern in three
// instructions: widen, add 1.5*2^52 (the integer lands in the low mantissa bits), strip the exponent
// pattern from the high word.  (A float -> int64 conversion proper is ~12 VALU instructions and the
// scatter does 16 of them per point.)
__device__ __forceinline__ unsigned long long fixed_from_float(float x) {
#ifdef ADV_CVT_I64
  return (unsigned long long)__float2ll_rn(x);
#else
  const double d = (double)x + 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
#endif
}

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
#ifdef ADV_IEEE_DIV
  const float gs = (st.s >= -CLAMP_HI && st.s <= CLAMP_HI) ? gphi_c / sqrtf(1.0f - sc * sc) : 0.f;
  const float den = st.n * st.n + st.d * st.d;
  const float gn = glam_c * st.d / den;
  const float gd = -glam_c * st.n / den;
#else
  // v_rsq / v_rcp (1 ulp) with one Newton step on the reciprocal: gradient error ~1e-7 relative
  const float gs = (st.s >= -CLAMP_HI && st.s <= CLAMP_HI) ? gphi_c * __builtin_amdgcn_rsqf(1.0f - sc * sc) : 0.f;
  const float den = st.n * st.n + st.d * st.d;
  float rden = __builtin_amdgcn_rcpf(den);
  rden = fmaf(fmaf(-den, rden, 1.0f), rden, rden);
  const float gn = glam_c * st.d * rden;
  const float gd = -glam_c * st.n * rden;
#endif
  const float gphi = gs * (st.cp * ca - st.sp * st.cl * sa) + gn * (-st.sp * st.sl) +
                     gd * (-st.sp * st.cl * ca - st.cp * sa);
  const float glam = gs * (-st.cp * st.sl * sa) + gn * (st.cp * st.cl) + gd * (-st.cp * st.sl * ca);
  gu = -dt * glam;
  gv = -dt * gphi;
}

template <int MODE, bool WHOLE, int NTH>
__global__ void __launch_bounds__(NTH)
sl_advect_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ field,
                     const float* __restrict__ u, const float* __restrict__ v,
                     float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                     const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lon, const float* __restrict__ fmeans,
                     const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                     int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, int halo, int tiles_x, int tiles,
                     int vec4) {
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
  float* misc = win + wn2;   // [0..1] pole means of gout, [2..2+NTH/64) per-wave max |cotangent|

  float m0 = 0.f, m1 = 0.f, gm0 = 0.f, gm1 = 0.f;
  if (!WHOLE) {
    m0 = fmeans[2 * plane]; m1 = fmeans[2 * plane + 1];
    gm0 = gmeans[2 * plane]; gm1 = gmeans[2 * plane + 1];
  }
  if (WHOLE && vec4) stage_plane_vec4(win, F, H, W, p);
  else stage_window(win, F, w, H, W, p, !WHOLE, m0, m1, NTH);
  for (int i = tid; i < wn; i += NTH) acc[i] = 0ull;
  // max |cotangent| over this workgroup's arrival points -> fixed-point scale
  const int npts = th * tw;
  float gmax = WHOLE ? 0.f : fmaxf(fabsf(gm0), fabsf(gm1));
  {
    TileIter itg(tid, tw, NTH);
    for (int i = tid; i < npts; i += NTH, itg.next())
      gmax = fmaxf(gmax, fabsf(GO[(ty0 + itg.yl) * W + tx0 + itg.xl]));
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
    __syncthreads();
    gm0 = misc[0]; gm1 = misc[1];
  }
  float scale, inv_scale;   // every thread derives the same power-of-two scale
  float mxall = misc[2];
#pragma unroll
  for (int q = 1; q < NTH / 64; ++q) mxall = fmaxf(mxall, misc[2 + q]);
  fixed_point_scale(mxall, scale, inv_scale);
  const bool rowwave = WHOLE && W == 64;

  const float kx = ((float)W - 1.0f) / g.d_lon, ky = ((float)H - 1.0f) / g.d_lat;
  TileIter it(tid, tw, NTH);
  // operands of point i+1 are in flight while point i is computed (see the forward kernel)
  float nu = 0.f, nv = 0.f, nsa = 0.f, nca = 0.f, nlo = 0.f, ngo = 0.f;
  if (tid < npts) {
    const int idx = (ty0 + it.yl) * W + tx0 + it.xl;
    nu = U[idx]; nv = V[idx]; nsa = sin_lat[idx]; nca = cos_lat[idx]; nlo = lon[idx]; ngo = GO[idx];
  }
  for (int i = tid; i < npts; i += NTH) {
    const int y = ty0 + it.yl, x = tx0 + it.xl, idx = y * W + x;
    const float cu = nu, cv = nv, sa = nsa, ca = nca, clo = nlo, cgo = ngo;
    it.next();
    if (i + NTH < npts) {
      const int nidx = (ty0 + it.yl) * W + tx0 + it.xl;
      nu = U[nidx]; nv = V[nidx]; nsa = sin_lat[nidx]; nca = cos_lat[nidx]; nlo = lon[nidx]; ngo = GO[nidx];
    }
    float ix, iy, tx, ty, wx[NT], wy[NT], dwx[NT], dwy[NT];
    int bx, by, sx, sy;
    DepState st;
    departure(cu, cv, sa, ca, clo, g, ix, iy, &st);
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    Interp<MODE>::dweights(tx, dwx);
    Interp<MODE>::dweights(ty, dwy);
    if (sx | sy) {
      shift_weights<NT>(wx, sx);
      shift_weights<NT>(dwx, sx);
      shift_weights<NT>(wy, sy);
      shift_weights<NT>(dwy, sy);
    }
    const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : cgo);
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
          atomicAdd(&acc[cell], fixed_from_float(gwy * wx[bb]));
#endif
          sxv = fmaf(val, wx[bb], sxv);
          sdx = fmaf(val, dwx[bb], sdx);
        }
        gix = fmaf(wy[a], sdx, gix);
        giy = fmaf(dwy[a], sxv, giy);
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
  const double inv = (double)inv_scale;
  if (WHOLE) {
    // fold the halo back: every source cell sums its aliases (adjoint of the a1 map), then the
    // adjoint of the first pole mean (rows 0, H-1 <- their mean)
    for (int i = tid; i < P; i += NTH) {
      const int y = i / W, x = i - y * W;
      long long s = 0;
      geo_for_each_alias(y, x, H, W, p, [&](int ii, int jj) { s += (long long)acc[(ii + p) * Wp + jj + p]; });
      float val = (float)((double)s * inv);
      if (rowwave) {   // a row is one wave's lanes in one iteration
        if (y == 0 || y == H - 1) val = wave_sum(val) / (float)W;
        GF[i] = val;
      } else {
        win[i] = val;  // the float plane reuses the window storage
      }
    }
    if (!rowwave) {
      __syncthreads();
      if (wave < 2) {
        float* row = win + (wave == 0 ? 0 : (H - 1) * W);
        const float m = wave_row_mean(row, W);
        for (int x = tid & 63; x < W; x += 64) row[x] = m;
      }
      __syncthreads();
      for (int i = tid; i < P; i += NTH) GF[i] = win[i];
    }
  } else {
    // flush the window once: one global float atomic per touched cell instead of 16 per point
    // (consecutive lanes -> consecutive cells: the 16 atomics per 64-byte line of one wave-instruction
    //  are combined by the memory pipeline; spreading them over 64 lines measured 2x slower)
    for (int i = tid; i < wn; i += NTH) {
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

AdvGeom make_geom(int H, int W, int p, float dt, float min_lat, float min_lon, float d_lat, float d_lon) {
  AdvGeom g{H, W, p, dt, min_lat, min_lon, d_lat, d_lon, 0.f, 0.f, 0.f, 0.f};
  g.r_lat = (float)(1.0 / (double)d_lat);
  g.r_lon = (float)(1.0 / (double)d_lon);
  g.r_wpm1 = (float)(1.0 / (double)(W + 2 * p - 1));
  g.r_hpm1 = (float)(1.0 / (double)(H + 2 * p - 1));
  return g;
}

constexpr size_t WHOLE_LDS_LIMIT = 64 * 1024;
// window halos (padded cells) of the tiled schedule.  Forward windows are cheap (4 B/cell); the
// backward holds 12 B/cell (64-bit accumulators + field), so its halo is what LDS allows at 2
// workgroups per CU.  Taps outside the window take the L2 / global-atomic path.
constexpr int HALO_FWD = 8;    // in-model optimum 6-12 at 128x256 and 721x1440; 24 pays only for ~45 px displacements
constexpr int HALO_BWD = 10;   // two workgroups of 512 threads per CU
constexpr int MAX_HALO = 32;

template <typename K>
int reserve_lds(K kernel, const char* what) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess) {
    paradis_set_error(what);
    return 2;
  }
  return 0;
}

// `flags` of the C ABI (include/paradis_hip.h, PARADIS_ADVECT_*): schedule choice and window halo are
// per-call arguments, the library keeps no mutable state
bool use_tiled(size_t whole_bytes, int flags) {
  const int sched = flags & PARADIS_ADVECT_SCHEDULE_MASK;
  if (sched == PARADIS_ADVECT_TILED) return true;
  return whole_bytes > WHOLE_LDS_LIMIT;
}
int halo_of(int flags, int dflt, bool backward) {
  int h = (flags >> PARADIS_ADVECT_HALO_SHIFT) & 0xff;          // 0 = default, else halo + 1
  const int hb = (flags >> PARADIS_ADVECT_HALO_BWD_SHIFT) & 0xff;
  if (backward && hb) h = hb;
  return h == 0 ? dflt : std::min(h - 1, MAX_HALO);
}

}  // namespace

extern "C" size_t paradis_sl_advect_ws_bytes(int B, int K, int H, int W) {
  (void)H; (void)W;
  return (size_t)B * K * 4 * sizeof(float) + 256;
}

#define ADV_LAUNCH(KERNEL, WHOLE_, NTH_, grid, lds, ...)                                                  \
  do {                                                                                                    \
    if (mode == PARADIS_INTERP_BICUBIC)                                                                   \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BICUBIC, WHOLE_, NTH_>), dim3(grid), dim3(NTH_), lds, st,  \
                         __VA_ARGS__);                                                                    \
    else                                                                                                  \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BILINEAR, WHOLE_, NTH_>), dim3(grid), dim3(NTH_), lds, st, \
                         __VA_ARGS__);                                                                    \
  } while (0)

extern "C" int paradis_sl_advect_fwd(const float* field, const float* u, const float* v, float* out,
                                     const float* sin_lat, const float* cos_lat, const float* lon,
                                     int B, int K, int H, int W, int64_t f_bs, int64_t uv_bs,
                                     int64_t o_bs, float dt, float min_lat, float min_lon,
                                     float d_lat, float d_lon, int mode, int flags, void* workspace,
                                     void* stream) {
  if (int e = check_adv("sl_advect_fwd", B, K, H, W, mode)) return e;
  if (B == 0) return 0;
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1, NT = 2 * p;
  AdvGeom g = make_geom(H, W, p, dt, min_lat, min_lon, d_lat, d_lon);
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K;
  const size_t whole = ((size_t)(H + 2 * p) * (W + 2 * p) + 2 * W) * sizeof(float);
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // 16-byte staging path: aligned planes, p even (bicubic), padded width even
  const int vec4 = (W % 4 == 0) && (f_bs % 4 == 0) && (((int64_t)H * W) % 4 == 0) && a16(field) && (p % 2 == 0);
  if (!use_tiled(whole, flags)) {
    ADV_LAUNCH(sl_advect_fwd_kernel, true, 256, planes, whole, field, u, v, out, sin_lat, cos_lat, lon,
               (const float*)nullptr, K, g, f_bs, uv_bs, o_bs, 0, 1, 1, vec4);
    PD_CHECK_LAUNCH("sl_advect_fwd");
    return 0;
  }
  PD_REQUIRE(workspace != nullptr, "sl_advect_fwd: workspace required for the tiled schedule");
  float* fmeans = (float*)workspace;
  const int mean_blocks = (planes * 2 * 64 + 255) / 256;
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, field, fmeans, planes, K, H, W, f_bs);
  const int tx = (W + TILE_W - 1) / TILE_W, ty = (H + TILE_HF - 1) / TILE_HF, tiles = tx * ty;
  PD_REQUIRE((int64_t)planes * tiles < (1ll << 31), "sl_advect_fwd: too many tiles");
  const int halo = halo_of(flags, HALO_FWD, false);
  const size_t lds = (size_t)(TILE_HF + 2 * halo + NT) * (TILE_W + 2 * halo + NT) * sizeof(float);
  PD_REQUIRE(lds <= 160 * 1024, "sl_advect_fwd: window does not fit LDS");
  static PerDeviceOnce once;
  if (once.first()) {
    if (reserve_lds(&sl_advect_fwd_kernel<PARADIS_INTERP_BICUBIC, false, TILED_THREADS_FWD>, "sl_advect_fwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_fwd_kernel<PARADIS_INTERP_BILINEAR, false, TILED_THREADS_FWD>, "sl_advect_fwd: cannot reserve LDS"))
      return 2;
  }
  ADV_LAUNCH(sl_advect_fwd_kernel, false, TILED_THREADS_FWD, (unsigned)(planes * tiles), lds, field, u, v, out, sin_lat,
             cos_lat, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, halo, tx, tiles, vec4);
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
                                     int flags, void* workspace, void* stream) {
  if (int e = check_adv("sl_advect_bwd", B, K, H, W, mode)) return e;
  if (B == 0) return 0;
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1, NT = 2 * p;
  AdvGeom g = make_geom(H, W, p, dt, min_lat, min_lon, d_lat, d_lon);
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K, P = H * W;
  auto lds_of = [](size_t cells) { return (3 * ((cells + 1) & ~(size_t)1) + 24) * sizeof(float); };
  const int vec4 = (W % 4 == 0) && (f_bs % 4 == 0) && (((int64_t)H * W) % 4 == 0) && (p % 2 == 0) &&
                   (reinterpret_cast<uintptr_t>(field) & 15) == 0;
  const size_t whole = lds_of((size_t)(H + 2 * p) * (W + 2 * p));
  if (!use_tiled(whole, flags)) {
    ADV_LAUNCH(sl_advect_bwd_kernel, true, 256, planes, whole, gout, field, u, v, gfield, gu, gv, sin_lat,
               cos_lat, lon, (const float*)nullptr, (const float*)nullptr, K, g, go_bs, f_bs, uv_bs,
               gf_bs, guv_bs, 0, 1, 1, vec4);
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
  const int halo = halo_of(flags, HALO_BWD, true);
  const size_t lds = lds_of((size_t)(TILE_H + 2 * halo + NT) * (TILE_W + 2 * halo + NT));
  PD_REQUIRE(lds <= 160 * 1024, "sl_advect_bwd: window does not fit LDS");
  static PerDeviceOnce once;
  if (once.first()) {
    if (reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BICUBIC, false, TILED_THREADS_BWD>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BILINEAR, false, TILED_THREADS_BWD>, "sl_advect_bwd: cannot reserve LDS"))
      return 2;
  }
  ADV_LAUNCH(sl_advect_bwd_kernel, false, TILED_THREADS_BWD, (unsigned)(planes * tiles), lds, gout, field, u, v, gfield,
             gu, gv, sin_lat, cos_lat, lon, (const float*)fmeans, (const float*)gmeans, K, g, go_bs,
             f_bs, uv_bs, gf_bs, guv_bs, halo, tx, tiles, vec4);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, gfield, planes, K, H, W, gf_bs);
  PD_CHECK_LAUNCH("sl_advect_bwd(tiled)");
  return 0;
}
