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
// The kernels are VALU-issue bound, not HBM bound (round 1: 264 issue units per point; on gfx950
// only fp32 add/mul/fma, v_mov, v_and and v_add_u32 issue at 2 cycles per wave, every compare, select,
// min/max/med3, conversion, shift, integer multiply-add and DPP move costs 4, transcendentals 8:
// tools/valu_rate_bench.hip).  Round 2 therefore restates the coordinate arithmetic:
//
//   * Sample coordinates.  The reference forms  lon_d = remainder(lon_a + atan2(n,d) + 2pi, 2pi),
//     pix = (lon_d - min)/d_lon (W-1), normalises to [-1,1] and ATen un-normalises again: ten fp32
//     roundings per axis.  Here  ix = wrap(lon_a cx + c0x + atan2(n,d) cx, [c0x, c0x + period))  and
//     iy = asin(s) cy + c0y  with cx = (W-1)/d_lon, cy = (H-1)/d_lat, period = 2 pi cx and the offsets
//     evaluated in double on the host (two FMAs and a floor per axis).  Same mathematical function;
//     its distance to the fp64 evaluation is HALF the CPU-fp32 reference's own (x: 1.4e-6 vs 3.4e-6
//     cells rms at 32x64), and its distance to the CPU-fp32 result is that reference's own rounding
//     noise (6.5e-6 rms at 32x64, tests/test_hip_pad_advect.py).
//   * asin / atan2 / sincos: polynomial kernels with wave-uniform range specialisation (a wave is one
//     latitude row: all its lanes sit in the same asin branch; displacements below 45 degrees need
//     neither range reduction nor octant fix-ups).  <= 2.3 ulp like the host libm.
//   * Cubic weights from A t (1-t)^2 and A t^2 (1-t) (5 instead of 8 operations for the outer taps).
//   * Tap origin: v_fract / v_med3 in float, one conversion for the LDS index.
//
// A workgroup stages a WINDOW of the geocyclic-PADDED plane in LDS through the a1 index map (pole rows
// already replaced by their means); the 4x4 / 2x2 taps of a point are plain 2-D indexing from one
// base address.  Three schedules of the same arithmetic:
//   row64 : W == 64, separable grid (the 5.625 degree configuration): window = the whole padded plane,
//           one workgroup per plane, one wave per latitude row: sin/cos(lat) are scalar loads, the
//           longitude column is a per-thread constant, pole-row means are wave reductions;
//   whole : any plane that fits LDS, per-point table loads;
//   tiled : 64x128 (forward) / 16x128 (backward) arrival tiles with a halo of D padded cells; taps
//           outside the window fall back to global loads through the index map / global float atomics.
// Backward scatters into 64-bit fixed-point window accumulators with integer LDS atomics
// (ds_add_f32 is ~30x slower than ds_add_u64 on gfx950, tools/lds_atomic_bench.hip), scaled by an
// exact power of two from the tile's max |cotangent| (resolution max|g| 2^-41; integer adds are
// associative: bitwise reproducible), then folds the halo back (whole) or flushes the window once
// with global atomics (tiled).
#include <stdlib.h>
#include <algorithm>
#include "common.h"

#pragma clang fp contract(off)   // every FMA below is explicit

namespace {

constexpr float CLAMP_HI = 0.9999999f;  // float(1 - 1e-7), as torch.clamp converts its python bound
constexpr float KA = -0.75f;
#ifndef ADV_TILE_H         // (A/B builds: tools/build_variant.sh)
#define ADV_TILE_H 16
#endif
#ifndef ADV_TILED_THREADS_BWD
#define ADV_TILED_THREADS_BWD 512
#endif
#ifndef ADV_HALO_BWD
#define ADV_HALO_BWD 10
#endif
constexpr int TILE_H = ADV_TILE_H, TILE_W = 128;  // arrival tile of the tiled schedule (backward)
// forward tile height: a taller tile amortises the halo (window cells per arrival point 2.6 at 16
// rows, 1.9 at 32, 1.5 at 64 with a halo of 8); the forward window is 4 B/cell, so LDS is not the
// limit.  Measured at 128x256: 1.64 / 1.42 / 1.23 ms per launch for 16 / 32 / 64 rows
constexpr int TILE_HF = 64;
// threads per tile in the tiled schedule: the window fixes the LDS per workgroup, so waves per SIMD
// come from the workgroup size.  Backward (12 B/cell, 2 workgroups per CU): at 256 threads it ran 1.7
// waves per SIMD at 29 % VALU issue, 512 threads measured 6.9 -> 5.6 ms at 128x256
constexpr int TILED_THREADS_FWD = 512, TILED_THREADS_BWD = ADV_TILED_THREADS_BWD;
constexpr int ADV_PF = 2;   // prefetch distance (points) of the operand loads

struct AdvGeom {
  int H, W, p;
  float ndt;              // -dt
  float cx, cy;           // cells per radian: (W-1)/d_lon, (H-1)/d_lat
  float per, inv_per;     // longitude period in cells (2 pi cx) and its reciprocal
  float c0x, c0y;         // p - min_lon cx,  p - min_lat cy
  float qoff;             // -c0x / period: the wrap is taken on [c0x, c0x + period)
  double c0xd;            // c0x in double: folded into the longitude table (lon_cells)
  double cxd;             // cx in double: lon -> cells conversion of the longitude table
};

struct DepState {  // intermediates needed by the backward chain
  float sp, cp, sl, cl, s, n, d;
};

// ---- elementary functions ------------------------------------------------------------------
// sin and cos: cephes minimax polynomials on [-pi/4, pi/4] (<= ~1 ulp) behind a Cody-Waite reduction
// (fdlibm's float split of pi/2); huge arguments take the ocml path.
__device__ __forceinline__ void sincos_kernel(float r, float& ps, float& pc) {
  const float z = r * r;
  ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
            fmaf(-0.5f, z, 1.0f));
}

__device__ __forceinline__ void sincos_reduced(float x, float& s, float& c) {
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
// per step - and needs no reduction and no quadrant selects: same bits, 10 instead of ~35 operations.
// |r| <= 1/8: the series two terms shorter, same error bounds (sin 0.51 ulp, cos 1.07 ulp)
__device__ __forceinline__ void sincos_kernel_small(float r, float& ps, float& pc) {
  const float z = r * r;
  ps = fmaf(fmaf(8.3333310e-3f, z, -1.6666667e-1f) * z, r, r);
  pc = fmaf(4.1666668e-2f * z, z, fmaf(-0.5f, z, 1.0f));
}

// Returns the tier taken (wave-uniform): 0 = every lane below 0.125 rad, 1 = below 0.78, 2 = general.
__device__ __forceinline__ int sincos_pair(float phi, float lam, float& sp, float& cp, float& sl, float& cl) {
  const float big = fmaxf(fabsf(phi), fabsf(lam));
  if (__all(big < 0.125f)) {     // displacements below 7 degrees per step: the usual case
    sincos_kernel_small(phi, sp, cp);
    sincos_kernel_small(lam, sl, cl);
    return 0;
  }
  if (__all(big < 0.78f)) {
    sincos_kernel(phi, sp, cp);
    sincos_kernel(lam, sl, cl);
    return 1;
  }
  sincos_reduced(phi, sp, cp);
  sincos_reduced(lam, sl, cl);
  return 2;
}

// (asin(x) - x) / x^3 as a polynomial in y = x^2 on [0, 1/4]; with the half-angle identity
// asin(x) = pi/2 - 2 asin(sqrt((1-x)/2)) for |x| >= 1/2 the result is within 2.3 ulp (mean 0.45),
// the host libm's float asin within 3.1 (mean 0.44)
__device__ __forceinline__ float asin_poly(float y) {
  float p = fmaf(0x1.15e14ep-5f, y, 0x1.169fe6p-6f);
  p = fmaf(p, y, 0x1.fe10a0p-6f);
  p = fmaf(p, y, 0x1.6d55e8p-5f);
  p = fmaf(p, y, 0x1.333448p-4f);
  return fmaf(p, y, 0x1.555554p-3f);
}

// asin for |x| < 1.  The lanes of a wave share a latitude row in every schedule, so the branch is
// wave-uniform almost always; the mixed case evaluates one polynomial behind selects.
__device__ __forceinline__ float asin_wave(float x) {
  const float ax = fabsf(x);
  const bool big = ax >= 0.5f;
  if (!__any(big)) {
    const float y = x * x;
    return fmaf(x, y * asin_poly(y), x);
  }
  const float t = fmaf(ax, -0.5f, 0.5f);
  const float r = __builtin_amdgcn_sqrtf(t);   // 1 ulp; contributes <= 0.5 ulp of the result
  float y = t, a = r;
  const bool mixed = !__all(big);
  if (mixed) {
    y = big ? t : ax * ax;
    a = big ? r : ax;
  }
  const float yy = fmaf(a, y * asin_poly(y), a);
  float res = fmaf(-2.0f, yy, 0x1.921fb6p+0f);
  if (mixed) res = big ? res : yy;
  return copysignf(res, x);
}

__device__ __forceinline__ float atan_poly(float t) {   // ocml's degree-8 minimax in t^2, |t| <= 1
  const float z = t * t;
  float pp = fmaf(z, 0x1.5a54bp-9f, -0x1.f4b218p-7f);
  pp = fmaf(z, pp, 0x1.53f67ep-5f);
  pp = fmaf(z, pp, -0x1.2fa9aep-4f);
  pp = fmaf(z, pp, 0x1.b26364p-4f);
  pp = fmaf(z, pp, -0x1.22c1ccp-3f);
  pp = fmaf(z, pp, 0x1.99717ep-3f);
  pp = fmaf(z, pp, -0x1.5554c4p-2f);
  return fmaf(t, z * pp, t);
}

// atan2 for finite arguments of ordinary magnitude (here y^2 + x^2 = cos^2(lat_d) > 1e-7).  When every
// lane of the wave has |y| < x (the departure point is less than 45 degrees of longitude away: the
// normal case) the quotient needs no octant bookkeeping.
__device__ __forceinline__ float atan2_wave(float y, float x) {
  if (__all(fabsf(y) < x)) return atan_poly(y * __builtin_amdgcn_rcpf(x));
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = atan_poly(mn * __builtin_amdgcn_rcpf(mx));
  a = (ay > ax) ? 0x1.921fb6p+0f - a : a;
  a = (x < 0.f) ? 0x1.921fb6p+1f - a : a;
  a = (y == 0.f) ? ((__float_as_int(x) < 0) ? 0x1.921fb6p+1f : 0.f) : a;   // also covers 0/0
  return copysignf(a, y);
}

// lonc = lon_a cx (cells), sa/ca = sin/cos(lat_a).  Returns the sample coordinates on the padded plane.
__device__ __forceinline__ void departure(float u, float v, float sa, float ca, float lonc,
                                          const AdvGeom& g, float& ix, float& iy, DepState* st) {
  const float lam = u * g.ndt;
  const float phi = v * g.ndt;
  float sp, cp, sl, cl;
  sincos_pair(phi, lam, sp, cp, sl, cl);
  const float cc = cp * cl;
  const float s = fmaf(sp, ca, cc * sa);
  const float sc = __builtin_amdgcn_fmed3f(s, -CLAMP_HI, CLAMP_HI);
  const float lat_d = asin_wave(sc);
  const float n = cp * sl;
  const float d = fmaf(cc, ca, -(sp * sa));
  const float a = atan2_wave(n, d);
  const float t = fmaf(a, g.cx, lonc);                 // unwrapped departure longitude in padded cells
  const float q = floorf(fmaf(t, g.inv_per, g.qoff));
  ix = fmaf(-q, g.per, t);                             // in [c0x, c0x + period) up to one rounding
  iy = fmaf(lat_d, g.cy, g.c0y);
  if (st) {
    st->sp = sp; st->cp = cp; st->sl = sl; st->cl = cl; st->s = s; st->n = n; st->d = d;
  }
}

// The map evaluated per lane, without the wave-uniform branches of the elementary functions above: the branch a point
// takes there depends on the other 63 points of its wave, which is harmless where the assignment of points to waves is
// fixed by the launch geometry, but not for the deferred points of the strip schedule - the order in which waves
// append to a strip's list varies from run to run, and with it a point's wave mates (a polar point amplifies the
// difference between two sin/cos tiers to 2e-4 of its velocity gradient).
__device__ __forceinline__ void departure_lane(float u, float v, float sa, float ca, float lonc,
                                               const AdvGeom& g, float& ix, float& iy, DepState* st) {
  const float lam = u * g.ndt;
  const float phi = v * g.ndt;
  float sp, cp, sl, cl;
  sincos_reduced(phi, sp, cp);
  sincos_reduced(lam, sl, cl);
  const float cc = cp * cl;
  const float s = fmaf(sp, ca, cc * sa);
  const float sc = __builtin_amdgcn_fmed3f(s, -CLAMP_HI, CLAMP_HI);
  // asin: one polynomial behind selects (the mixed case of asin_wave)
  float lat_d;
  {
    const float ax = fabsf(sc);
    const bool big = ax >= 0.5f;
    const float t = fmaf(ax, -0.5f, 0.5f);
    const float r = __builtin_amdgcn_sqrtf(t);
    const float y = big ? t : ax * ax;
    const float a = big ? r : ax;
    const float yy = fmaf(a, y * asin_poly(y), a);
    lat_d = copysignf(big ? fmaf(-2.0f, yy, 0x1.921fb6p+0f) : yy, sc);
  }
  const float n = cp * sl;
  const float d = fmaf(cc, ca, -(sp * sa));
  float a;
  {   // the octant form of atan2_wave
    const float ax = fabsf(d), ay = fabsf(n);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    a = atan_poly(mn * __builtin_amdgcn_rcpf(mx));
    a = (ay > ax) ? 0x1.921fb6p+0f - a : a;
    a = (d < 0.f) ? 0x1.921fb6p+1f - a : a;
    a = (n == 0.f) ? ((__float_as_int(d) < 0) ? 0x1.921fb6p+1f : 0.f) : a;
    a = copysignf(a, n);
  }
  const float t = fmaf(a, g.cx, lonc);
  const float q = floorf(fmaf(t, g.inv_per, g.qoff));
  ix = fmaf(-q, g.per, t);
  iy = fmaf(lat_d, g.cy, g.c0y);
  if (st) {
    st->sp = sp; st->cp = cp; st->sl = sl; st->cl = cl; st->s = s; st->n = n; st->d = d;
  }
}

// The same map for a wave that is ONE LATITUDE ROW (separable schedules): sa, ca and iya - the arrival latitude
// in padded cells, p + (lat_a - min_lat) cy - are wave-uniform scalars.  The small-displacement regime is decided
// ONCE per row by a single vector condition - both rotation angles below 0.125 rad, departure less than 45 degrees
// of longitude away, no clamp active - behind which the code is straight-line: short sin/cos series, plain atan
// quotient, and the latitude either from the short asin polynomial (rows within asin's small-argument range for
// every such displacement: |sin(lat_a)| < 1/4) or RELATIVE to the arrival latitude:
//     sin(lat_d - lat_a) = s cos(lat_a) - cos(lat_d) sin(lat_a) = sin(phi') - sin(lat_a) (cos(lat_d) - d),
//     cos(lat_d) = sqrt(n^2 + d^2)
// (expand s and d: the products of sa, ca cancel exactly), iy = iya + cy asin(small argument): no half-angle chain,
// and the rounding of s is not amplified by 1 / cos(lat_d) next to the poles - against the fp64 evaluation this form
// is ~20 x closer than asin(s) in fp32 at every grid size.  Any other wave takes departure().
__device__ __forceinline__ void departure_row(float u, float v, float sa, float ca, float lonc, float iya,
                                              const AdvGeom& g, float& ix, float& iy, DepState* st) {
  const float lam = u * g.ndt;
  const float phi = v * g.ndt;
  float sp, cp, sl, cl;
  sincos_kernel_small(phi, sp, cp);
  sincos_kernel_small(lam, sl, cl);
  const float cc = cp * cl;
  const float s = fmaf(sp, ca, cc * sa);
  const float n = cp * sl;
  const float d = fmaf(cc, ca, -(sp * sa));
#ifndef ADV_NO_REL    // (diagnostic A/B builds only)
  const bool ok = fmaxf(fabsf(phi), fabsf(lam)) < 0.125f && fabsf(n) < d && fabsf(s) <= CLAMP_HI;
#else
  const bool ok = false;
#endif
  if (!__all(ok)) {
    departure(u, v, sa, ca, lonc, g, ix, iy, st);
    return;
  }
  if (fabsf(sa) < 0.25f) {           // scalar: |s| <= |sa| + sin(0.25) < 1/2 for every lane
    const float y2 = s * s;
    iy = fmaf(fmaf(s, y2 * asin_poly(y2), s), g.cy, g.c0y);
  } else {
    const float cosd = __builtin_amdgcn_sqrtf(fmaf(d, d, n * n));
    const float arg = fmaf(-sa, cosd - d, sp);         // |arg| <= sin(0.25)
    const float y2 = arg * arg;
    iy = fmaf(fmaf(arg, y2 * asin_poly(y2), arg), g.cy, iya);
  }
  const float a = atan_poly(n * __builtin_amdgcn_rcpf(d));
  const float t = fmaf(a, g.cx, lonc);
  const float q = floorf(fmaf(t, g.inv_per, g.qoff));
  ix = fmaf(-q, g.per, t);
  if (st) {
    st->sp = sp; st->cp = cp; st->sl = sl; st->cl = cl; st->s = s; st->n = n; st->d = d;
  }
}

// A wave-uniform pointer pinned to scalar registers: `srow(p)[lane]` with an unsigned 32-bit lane then
// becomes a global access with a scalar base and a 32-bit vector offset (no 64-bit vector address
// arithmetic per load: 7 half-rate VALU operations per access in the first version of these loops).
template <typename T>
using global_ptr = __attribute__((address_space(1))) T*;   // explicit: an integer-built pointer would be `flat`
template <typename T>
__device__ __forceinline__ global_ptr<T> srow(T* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (global_ptr<T>)(((uint64_t)hi << 32) | lo);
}

// arrival longitude in padded cells: lon cx + c0x, one rounding
__device__ __forceinline__ float lon_cells(float lon, const AdvGeom& g) { return (float)fma((double)lon, g.cxd, g.c0xd); }

// ---- interpolation weights -------------------------------------------------------------------
// (the weights and tap sums are not coordinate-critical - an ulp of a weight is 1e-7 relative in the
//  result, whereas an ulp of a sample coordinate is multiplied by the field slope)
__device__ __forceinline__ float cub1(float x) { return fmaf(fmaf(KA + 2.f, x, -(KA + 3.f)) * x, x, 1.f); }
__device__ __forceinline__ float dcub1(float x) { return fmaf(3.f * (KA + 2.f), x, -2.f * (KA + 3.f)) * x; }

template <int MODE>
struct Interp {
  static constexpr int NT = (MODE == PARADIS_INTERP_BICUBIC) ? 4 : 2;
  static constexpr int OFF0 = (MODE == PARADIS_INTERP_BICUBIC) ? -1 : 0;
  // Keys cubic convolution, A = -0.75: taps at -1, 0, 1, 2 carry  A t (1-t)^2,  c1(t),  c1(1-t),  A t^2 (1-t)
  static __device__ __forceinline__ void weights(float t, float* w) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      // the four weights sum to 1 and w0 + w3 = A t (1-t) (t + (1-t)): the fourth costs two subtractions
      const float um = 1.f - t, atu = (KA * t) * um;
      w[0] = atu * um; w[1] = cub1(t); w[3] = atu * t; w[2] = (1.f - w[1]) - atu;
    } else {
      w[0] = 1.f - t; w[1] = t;
    }
  }
  static __device__ __forceinline__ void dweights(float t, float* dw) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      const float um = 1.f - t;
      dw[0] = fmaf(fmaf(3.f * KA, t, -4.f * KA), t, KA);     // A (3t^2 - 4t + 1)
      dw[1] = dcub1(t);
      dw[2] = -dcub1(um);
      dw[3] = fmaf(-3.f * KA, t, 2.f * KA) * t;              // A (2t - 3t^2)
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

// Tap block of a point when the window is the WHOLE padded plane: fraction, clamped origin and the
// LDS index in float arithmetic (exact: every value is an integer below 2^24), one conversion.
// Returns true when the clamp moved the origin (a coordinate on the plane edge, or not finite): the
// caller takes the general path.  `cell` indexes tap (0,0) BEFORE the OFF0 shift.
template <int MODE>
__device__ __forceinline__ bool tap_block_whole(float ix, float iy, float Hpf, float Wpf, float& tx, float& ty,
                                                int& cell) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  tx = __builtin_amdgcn_fractf(ix);
  ty = __builtin_amdgcn_fractf(iy);
  const float x0f = ix - tx, y0f = iy - ty;
  const float xc = __builtin_amdgcn_fmed3f(x0f, (float)(-OFF0), Wpf - (float)(NT + OFF0));
  const float yc = __builtin_amdgcn_fmed3f(y0f, (float)(-OFF0), Hpf - (float)(NT + OFF0));
  cell = (int)fmaf(yc, Wpf, xc);
  return !(xc == x0f && yc == y0f);
}

__device__ __forceinline__ float wave_row_mean(const float* row, int W) {
  float s = 0.f;
  for (int x = threadIdx.x & 63; x < W; x += 64) s += row[x];
  return wave_sum_dpp(s) / (float)W;
}

// Window of the padded plane held in LDS: padded rows [wy0, wy0+WH), padded cols [wx0, wx0+WW).
struct Window {
  int wy0, wx0, WH, WW;
};

// iterate i = tid, tid+nth, ... < th*tw as (yl, xl) without a division per point
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
// rows 0 / H-1 by the given means (tiled schedule; whole-plane schedules compute the means in LDS).
// Flat over the window in batches: all loads of a batch are issued before the first LDS write, so a
// workgroup pays ~one memory round trip for its window.
constexpr int STAGE_BATCH = 6;
__device__ __forceinline__ void stage_window(float* win, const float* __restrict__ F, const Window& w,
                                             int H, int W, int p, bool subst, float m0, float m1,
                                             int nth = 256) {
  // A thread keeps one window column (its longitude wrap - plain and mirrored - is computed once) and
  // walks down the rows; only the cheap row map (mirror beyond a pole) is per element.
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

// pole rows of a whole padded plane in LDS <- their mean over the W interior columns (lon halo included)
__device__ __forceinline__ void pole_rows_to_mean_lds(float* win, int H, int W, int p, int Wp) {
  const int wave = threadIdx.x >> 6;
  if (wave < 2) {
    float* row = win + (wave == 0 ? p : H - 1 + p) * Wp;
    const float m = wave_row_mean(row + p, W);
    for (int x = threadIdx.x & 63; x < Wp; x += 64) row[x] = m;
  }
}

// ======================================================================================
// forward
// ======================================================================================
// value of one arrival point from a WHOLE-plane window
template <int MODE>
__device__ __forceinline__ float sample_whole(const float* win, float ix, float iy, int Hp, int Wp, float Hpf,
                                              float Wpf) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  float tx, ty, wx[NT], wy[NT];
  int cell;
  const bool edge = tap_block_whole<MODE>(ix, iy, Hpf, Wpf, tx, ty, cell);
  const float* base = win + OFF0 * (Wp + 1) + cell;
  if (__any(edge)) {   // a coordinate rounded onto the plane edge, or is not finite: general origin + shifted weights
    int bx, by, sx, sy;
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    shift_weights<NT>(wx, sx);
    shift_weights<NT>(wy, sy);
    base = win + by * Wp + bx;
  } else {
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
  }
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float rowacc = 0.f;
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * Wp + bb], wx[bb], rowacc);
    acc = fmaf(rowacc, wy[a], acc);
  }
  return acc;
}

// Value of one arrival point from a whole-plane window that is XR = 1 column wider than the padded plane
// (column Wp repeats the wrap): for finite inputs the tap block then always lies inside the window -
// ix is in [p - eps, W + p + eps], iy within half a cell of the grid's latitude range - so the forward
// needs neither clamps nor the edge path.  A non-finite coordinate makes every weight NaN and a garbage
// index, and LDS reads beyond the allocation return 0: the result is NaN, as it should be.
constexpr int ROW64_XR = 1;
// velocities and output are touched once per launch: non-temporal
#define ADV_LD(p) __builtin_nontemporal_load(p)
#define ADV_ST(v, p) __builtin_nontemporal_store(v, p)
template <int MODE>
__device__ __forceinline__ float sample_wide(const float* win, float ix, float iy, int WS, float WSf) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  const float tx = __builtin_amdgcn_fractf(ix), ty = __builtin_amdgcn_fractf(iy);
  float wx[NT], wy[NT];
  Interp<MODE>::weights(tx, wx);
  Interp<MODE>::weights(ty, wy);
  const int cell = (int)fmaf(iy - ty, WSf, ix - tx);
  const float* base = win + OFF0 * (WS + 1) + cell;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float rowacc = 0.f;
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * WS + bb], wx[bb], rowacc);
    acc = fmaf(rowacc, wy[a], acc);
  }
  return acc;
}

// W == 64, separable grid: a workgroup (4 waves) walks ROW64_CHUNK consecutive planes, wave w owns rows
// w, w+4, ... of each.  The planes are software pipelined through two LDS windows:
//   - the interior of plane n+1 goes global -> LDS by DMA (one 256-byte row per global_load_lds_dword)
//     while plane n is computed: no staging registers, no staging latency on the critical path;
//   - halo columns and the mirrored rows beyond the poles are copies of interior cells, filled LDS -> LDS;
//   - the velocity prefetch runs ADV_PF rows ahead ACROSS plane boundaries.
// XR = ROW64_XR: global grid (the host checked the coordinate range), wide window, unclamped taps;
// XR = 0: any other grid, taps outside the padded plane count as zero like ATen's grid_sample.
#ifndef ADV_ROW64_CHUNK     // (A/B builds)
#define ADV_ROW64_CHUNK 4
#endif
constexpr int ROW64_CHUNK = ADV_ROW64_CHUNK;   // 1 and 12 (one workgroup per resident slot) measured 3-6 % slower
typedef __attribute__((address_space(3))) void* adv_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* adv_gbl_ptr_t;

// velocity prefetch cursor of a wave: ADV_PF rows ahead of the row being computed, across plane boundaries.
// Element offsets, not pointers: the loads must stay `global` with a scalar base (srow).
struct VelCursor {
  int64_t off, uv_bs;      // offset of the cursor's plane in u and in v
  int plane, last, b, k, K, y, wave, H, P;
  int yend;                // wave + 4 * (row slots per plane): ceil(H / 4) rounded up to a multiple of 2 ADV_PF
  __device__ __forceinline__ void load(const float* __restrict__ u, const float* __restrict__ v, unsigned lane,
                                       float& a, float& c) {
    const int64_t j = off + min(y, H - 1) * 64;
    a = ADV_LD(&srow(u + j)[lane]);
    c = ADV_LD(&srow(v + j)[lane]);
    y += 4;
    if (y >= yend && plane + 1 < last) {   // (past the last plane: keeps reloading its last row)
      ++plane; y = wave;
      if (++k == K) { k = 0; ++b; }
      off = (int64_t)b * uv_bs + (int64_t)k * P;
    }
  }
};

// One 256-byte row global -> LDS by DMA (M0 = LDS row start, lane -> +4 bytes), both addresses wave-uniform.
// Inline assembly on purpose: with the builtin the compiler's wait-count pass treats the vector memory
// counter as unordered while a DMA is pending and turns every wait for a prefetched velocity into vmcnt(0).
// Hidden from it, its counted waits only become stricter (they count fewer newer operations than there are).
__device__ __forceinline__ void dma_row_to_lds(const float* grow, unsigned lane_bytes, float* lds_row) {
  const uint64_t a = (uint64_t)grow;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const uint64_t base = ((uint64_t)hi << 32) | lo;
  const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(adv_lds_ptr_t)lds_row);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
               :: "s"(m), "v"(lane_bytes), "s"(base) : "memory");   // (M0 is reserved: the compiler keeps nothing in it across statements)
}

// Halo cells of the W = 64 window: the 2p mirrored rows beyond the poles, then p + (p + XR) columns per row.  Cell q of
// that list as (destination << 16 | source) offsets into the window, 0xffff0000 for the cells of the two pole rows
// (written whole by pole_rows_to_mean_lds).  The list does not depend on the plane: each thread keeps its ROW64_HMAP
// pairs in registers for the whole launch instead of redoing the index arithmetic (~50 vector instructions) per plane.
constexpr int ROW64_HMAP = 3;
constexpr unsigned ROW64_NO_CELL = 0xffff0000u;   // (its source half reads cell 0)
template <int MODE, int XR>
__device__ __forceinline__ int row64_halo_cells(int H) {
  constexpr int p = Interp<MODE>::NT / 2, WS = 64 + 2 * p + XR;
  return 2 * p * WS + H * (2 * p + XR);
}
template <int MODE, int XR>
__device__ __forceinline__ unsigned row64_halo_pair(int q, int H) {
  constexpr int W = 64, p = Interp<MODE>::NT / 2, WS = W + 2 * p + XR, hc = 2 * p + XR, nrow_cells = 2 * p * WS;
  int lr, lc;
  if (q < nrow_cells) {
    const int rr = q / WS;
    lc = q - rr * WS;
    lr = rr < p ? rr : H + rr;
  } else {
    const int e = q - nrow_cells, rr = e / hc, cc = e - rr * hc;
    lr = rr + p;
    lc = cc < p ? cc : W + cc;
  }
  if (lr == p || lr == H - 1 + p) return ROW64_NO_CELL;
  int sr, sc;
  geo_src(lr - p, lc - p, H, W, sr, sc);
  return ((unsigned)(lr * WS + lc) << 16) | (unsigned)((sr + p) * WS + sc + p);
}

// One plane of the pipeline.  The interior of the NEXT plane goes global -> LDS by DMA, one row per row
// iteration, while this plane is computed.  That the DMA INTO cur - issued one plane earlier - has landed
// is the caller's counted wait.
template <int MODE, int XR>
__device__ __forceinline__ void row64_plane(float* __restrict__ cur, float* __restrict__ nxt,
                                            const float* __restrict__ field, int64_t next_off, bool has_next,
                                            float* __restrict__ O, const float* __restrict__ u,
                                            const float* __restrict__ v, VelCursor& vc, float (&qu)[ADV_PF],
                                            float (&qv)[ADV_PF], const float* __restrict__ sin_lat,
                                            const float* __restrict__ cos_lat, const float* __restrict__ lat_cells,
                                            float lonc, const AdvGeom& g, int wave, unsigned lane, bool fill_halo,
                                            const unsigned (&hmap)[ROW64_HMAP]) {
  constexpr int W = 64, p = Interp<MODE>::NT / 2, WS = W + 2 * p + XR;
  const int H = g.H, Hp = H + 2 * p, tid = threadIdx.x;
  // halo columns (p left, p + XR right) and the p mirrored rows beyond each pole are copies of interior
  // cells; the two pole rows are written whole by pole_rows_to_mean_lds
  if (fill_halo) {
    if (row64_halo_cells<MODE, XR>(H) <= 256 * ROW64_HMAP) {   // the (destination, source) cells of this thread: row64_halo_map
      float t[ROW64_HMAP];
#pragma unroll
      for (int k = 0; k < ROW64_HMAP; ++k) t[k] = cur[hmap[k] & 0xffffu];
#pragma unroll
      for (int k = 0; k < ROW64_HMAP; ++k)
        if (hmap[k] != ROW64_NO_CELL) cur[hmap[k] >> 16] = t[k];
    } else {
      for (int q = tid; q < row64_halo_cells<MODE, XR>(H); q += 256) {
        const unsigned m = row64_halo_pair<MODE, XR>(q, H);
        if (m != ROW64_NO_CELL) cur[m >> 16] = cur[m & 0xffffu];
      }
    }
  }
  pole_rows_to_mean_lds(cur, H, W, p, WS);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (no vmcnt: loads stay in flight)
  const float Hpf = (float)Hp, WSf = (float)WS;
  // one row: consume the velocities in (cu, cv), refill the slot (nu, nv) from the cursor
  auto do_row = [&](int y, float cu, float cv, float& nu, float& nv) {
    if (has_next && y < H) dma_row_to_lds(field + next_off + y * W, 4 * lane, nxt + (y + p) * WS + p);
    vc.load(u, v, lane, nu, nv);
    if (y < H) {
      const float sa = sin_lat[y * W], ca = cos_lat[y * W];   // uniform address: scalar loads
      float ix, iy;
      departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, nullptr);
      float acc = XR ? sample_wide<MODE>(cur, ix, iy, WS, WSf) : sample_whole<MODE>(cur, ix, iy, Hp, WS, Hpf, WSf);
      if (y == 0 || y == H - 1) acc = wave_sum_dpp(acc) * (1.0f / 64.0f);   // pole rows <- their mean
      ADV_ST(acc, &srow(O + y * W)[lane]);
    }
  };
  // Two register sets that swap roles (qu/qv -> ru/rv -> qu/qv): a load never targets a register whose old
  // value is still needed, so there is no copy of a just-loaded register - and no vmcnt(0) - at the back edge.
  float ru[ADV_PF], rv[ADV_PF];
  for (int y0 = wave; y0 < vc.yend; y0 += 8 * ADV_PF) {   // the cursor's slot count: the same for every wave
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) do_row(y0 + 4 * d, qu[d], qv[d], ru[d], rv[d]);
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) do_row(y0 + 4 * (ADV_PF + d), ru[d], rv[d], qu[d], qv[d]);
  }
}

// Scalar-register cap of the forward kernel.  The compiler's own choice (106 SGPRs: row pointers, table values and the
// constants of five code paths) admits 6 workgroups per CU (800 SGPRs per SIMD / (112 + 16)); at <= 80 the
// hardware admits 8, which is also what the 59 VGPRs and the 19.9 KB of LDS allow.  27 values then live in VGPR
// lanes (v_readlane / v_writelane around the rare paths): 178 -> 173 us per launch, same box (round 3).
#ifndef ADV_FWD_SGPRS      // (0 = the compiler's choice)
#define ADV_FWD_SGPRS 80
#endif
#if ADV_FWD_SGPRS
#define ADV_FWD_SGPR_ATTR __attribute__((amdgpu_num_sgpr(ADV_FWD_SGPRS)))
#else
#define ADV_FWD_SGPR_ATTR
#endif
template <int MODE, int XR>
__global__ void __launch_bounds__(256) ADV_FWD_SGPR_ATTR
sl_advect_fwd_row64(const float* __restrict__ field, const float* __restrict__ u,
                    const float* __restrict__ v, float* __restrict__ out,
                    const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                    const float* __restrict__ lat_cells, const float* __restrict__ lon, int K, AdvGeom g, int64_t f_bs, int64_t uv_bs,
                    int64_t o_bs, int planes, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int W = 64, p = Interp<MODE>::NT / 2, WS = W + 2 * p + XR;   // WS: window row stride
  const int H = g.H, P = H * W, Hp = H + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;   // unsigned: row pointer (scalar) + 32-bit lane offset addressing
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = blockIdx.x * chunk, last = min(first + chunk, planes);
  int b = first / K, k = first - b * K;

  VelCursor vc;
  vc.uv_bs = uv_bs; vc.plane = first; vc.last = last; vc.b = b; vc.k = k; vc.K = K;
  vc.y = wave; vc.wave = wave; vc.H = H; vc.P = P;
  vc.yend = wave + 4 * (((H + 3) / 4 + 2 * ADV_PF - 1) / (2 * ADV_PF) * (2 * ADV_PF));
  vc.off = (int64_t)b * uv_bs + (int64_t)k * P;
  float qu[ADV_PF], qv[ADV_PF];
#pragma unroll
  for (int d = 0; d < ADV_PF; ++d) vc.load(u, v, lane, qu[d], qv[d]);
  const float lonc = lon_cells(lon[lane], g);
  unsigned hmap[ROW64_HMAP];
#pragma unroll
  for (int k = 0; k < ROW64_HMAP; ++k) {
    const int q = tid + 256 * k;
    hmap[k] = q < row64_halo_cells<MODE, XR>(H) ? row64_halo_pair<MODE, XR>(q, H) : ROW64_NO_CELL;
  }
  float* cur = smem;
  float* nxt = smem + Hp * WS;
  {   // the first plane of the chunk is staged through registers, halo included
    Window w{0, 0, Hp, WS};
    stage_window(cur, field + (int64_t)b * f_bs + (int64_t)k * P, w, H, W, p, false, 0.f, 0.f, 256);
  }
  for (int plane = first; plane < last; ++plane) {
    // `cur` has landed and every wave is done reading `nxt`.  A wave issues its last DMA at the top of its
    // last row; two loads and one store (at least) follow.  Loads complete in issue order, so with at most
    // 2 operations outstanding the DMA - older than both loads - is in LDS whatever the store did; the
    // barrier covers the other waves' rows.
    if (plane != first) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int nb = b, nk = k + 1;
    if (nk == K) { nk = 0; ++nb; }
    const bool has_next = plane + 1 < last;
    const int64_t next_off = has_next ? (int64_t)nb * f_bs + (int64_t)nk * P : 0;
    row64_plane<MODE, XR>(cur, nxt, field, next_off, has_next, out + (int64_t)b * o_bs + (int64_t)k * P, u, v, vc,
                          qu, qv, sin_lat, cos_lat, lat_cells, lonc, g, wave, lane, plane != first, hmap);
    float* t = cur; cur = nxt; nxt = t;
    b = nb; k = nk;
  }
}

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
  float* win = smem;                           // [WH*WW]
  float* pole_out = win + w.WH * w.WW;         // [2*W]  (WHOLE only)

  float m0 = 0.f, m1 = 0.f;
  if (!WHOLE) { m0 = fmeans[2 * plane]; m1 = fmeans[2 * plane + 1]; }
  if (WHOLE && vec4) stage_plane_vec4(win, F, H, W, p);
  else stage_window(win, F, w, H, W, p, !WHOLE, m0, m1, NTH);
  __syncthreads();
  if (WHOLE) {
    pole_rows_to_mean_lds(win, H, W, p, Wp);
    __syncthreads();
  }
  const float Hpf = (float)Hp, Wpf = (float)Wp;
  const int npts = th * tw;
  // one arrival point of the tiled schedule: departure -> tap block -> window gather or L2 fallback
  auto point_tiled = [&](float uu, float vv, float sa, float ca, float lo) -> float {
    float ix, iy, tx, ty, wx[NT], wy[NT];
    int bx, by, sx, sy;
    departure(uu, vv, sa, ca, lon_cells(lo, g), g, ix, iy, nullptr);
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    if (sx | sy) {   // only when a coordinate rounds onto the plane edge (or is not finite)
      shift_weights<NT>(wx, sx);
      shift_weights<NT>(wy, sy);
    }
    int ry = by - w.wy0, rx = bx - w.wx0;
    if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
    const bool inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    float acc = 0.f;
    if (inwin) {
      const float* base = win + ry * w.WW + rx;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * w.WW + bb], wx[bb], rowacc);
        acc = fmaf(rowacc, wy[a], acc);
      }
    } else {  // taps served by L2 through the index map
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
          rowacc = fmaf(val, wx[bb], rowacc);
        }
        acc = fmaf(rowacc, wy[a], acc);
      }
    }
    return acc;
  };
  if constexpr (WHOLE) {
    // Whole plane: the arrival index is the flat index.  Operands are prefetched ADV_PF points ahead
    // into a queue whose slots are fixed registers (the loop is unrolled ADV_PF times) with
    // unconditional, clamped loads: straight-line code lets the compiler count vmcnt exactly.
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
          qu[d] = U[j]; qv[d] = V[j]; qs[d] = sin_lat[j]; qc[d] = cos_lat[j]; ql[d] = lon[j];
        }
        float ix, iy;
        departure(cu, cv, csa, cca, lon_cells(clo, g), g, ix, iy, nullptr);
        const float acc = sample_whole<MODE>(win, ix, iy, Hp, Wp, Hpf, Wpf);
        if (i < npts) {
          if (i < W || i >= lastrow0) pole_out[i < W ? i : W + i - lastrow0] = acc;
          else O[i] = acc;
        }
      }
    }
    __syncthreads();
    if (wave < 2) {
      const float* row = pole_out + (wave == 0 ? 0 : W);
      const float m = wave_row_mean(row, W);
      float* orow = O + (wave == 0 ? 0 : (int64_t)(H - 1) * W);
      for (int x = tid & 63; x < W; x += 64) orow[x] = m;
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
      const float acc = point_tiled(cu, cv, csa, cca, clo);
      O[idx] = acc;
    }
  }
}


// Tile geometry shared by the two row-wave tiled kernels: waves 2r and 2r+1 own the two 64-column halves
// of tile rows r, r + NW/2, ...; lanes beyond a ragged tile edge load clamped addresses and store nothing.
struct TileRowLane {
  int rs, rstep;        // first tile row of this wave, row step
  int cbase;            // image column of lane 0 (clamped inside the plane)
  unsigned lx;          // lane offset (clamped inside the tile)
  bool active;
  __device__ __forceinline__ TileRowLane(int wave, int nwaves, unsigned lane, int tx0, int tw, int W) {
    const int ch = wave & 1;
    rs = wave >> 1; rstep = nwaves >> 1;
    const int first = ch * 64;
    active = first + (int)lane < tw;
    cbase = min(tx0 + first, W - 1);
    const int last = max(tw - 1 - first, 0);
    lx = min(lane, (unsigned)last);
  }
};

// window-relative tap block of a point in float arithmetic (every value an integer below 2^24): returns
// true when the NT x NT block lies inside the window AND the plane clamp did not move it; `cell` = index
// of tap (0,0) in the window
template <int MODE>
__device__ __forceinline__ bool tap_block_window(float ix, float iy, float Hpf, float Wpf, float Wf, float wx0f,
                                                 float wy0f, float WWf, float WHf, float& tx, float& ty, int& cell) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  tx = __builtin_amdgcn_fractf(ix);
  ty = __builtin_amdgcn_fractf(iy);
  const float x0f = ix - tx, y0f = iy - ty;
  const float xc = __builtin_amdgcn_fmed3f(x0f, (float)(-OFF0), Wpf - (float)(NT + OFF0));
  const float yc = __builtin_amdgcn_fmed3f(y0f, (float)(-OFF0), Hpf - (float)(NT + OFF0));
  float rx = (xc + (float)OFF0) - wx0f;
  const float ry = (yc + (float)OFF0) - wy0f;
  rx = rx < 0.f ? rx + Wf : (rx > WWf - (float)NT ? rx - Wf : rx);   // the window may straddle the date line
  const float rxc = __builtin_amdgcn_fmed3f(rx, 0.f, WWf - (float)NT);
  const float ryc = __builtin_amdgcn_fmed3f(ry, 0.f, WHf - (float)NT);
  cell = (int)fmaf(ryc, WWf, rxc);
  return xc == x0f && yc == y0f && rxc == rx && ryc == ry;
}

// Tiled schedule on a separable grid, one wave per 64 columns of a tile row (scalar sin/cos(lat) loads,
// per-lane longitude, scalar row pointers, float tap blocks): same arithmetic as the row64 kernel, the
// window and the L2 fallback of the generic tiled kernel.
template <int MODE>
__global__ void __launch_bounds__(TILED_THREADS_FWD)
sl_advect_fwd_tilerow(const float* __restrict__ field, const float* __restrict__ u,
                      const float* __restrict__ v, float* __restrict__ out,
                      const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                      const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans, int K,
                      AdvGeom g, int64_t f_bs, int64_t uv_bs, int64_t o_bs, int halo, int tiles_x, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = Interp<MODE>::NT, NTH = TILED_THREADS_FWD;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x / tiles, tile = blockIdx.x - plane * tiles;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;
  const int ty0 = (tile / tiles_x) * TILE_HF, tx0 = (tile % tiles_x) * TILE_W;
  const int th = min(TILE_HF, H - ty0), tw = min(TILE_W, W - tx0);
  Window w{ty0 + p - halo, tx0 + p - halo, TILE_HF + 2 * halo + NT, TILE_W + 2 * halo + NT};
  float* win = smem;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const TileRowLane tl(wave, NTH / 64, lane, tx0, tw, W);

  float qu[ADV_PF], qv[ADV_PF];
#pragma unroll
  for (int d = 0; d < ADV_PF; ++d) {
    const int j = (ty0 + min(tl.rs + tl.rstep * d, th - 1)) * W + tl.cbase;
    qu[d] = srow(U + j)[tl.lx]; qv[d] = srow(V + j)[tl.lx];
  }
  const float lonc = lon_cells(srow(lon + tl.cbase)[tl.lx], g);
  stage_window(win, F, w, H, W, p, true, m0, m1, NTH);
  __syncthreads();

  const float Hpf = (float)Hp, Wpf = (float)Wp, Wf = (float)W, wx0f = (float)w.wx0, wy0f = (float)w.wy0,
              WWf = (float)w.WW, WHf = (float)w.WH;
  // general path of a point: any tap origin, taps from the window or through the index map from L2
  auto slow_point = [&](float ix, float iy) -> float {
    float tx, ty, wx[NT], wy[NT];
    int bx, by, sx, sy;
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx);
    Interp<MODE>::weights(ty, wy);
    shift_weights<NT>(wx, sx);
    shift_weights<NT>(wy, sy);
    int ry = by - w.wy0, rx = bx - w.wx0;
    if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
    const bool inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    float acc = 0.f;
    if (inwin) {
      const float* base = win + ry * w.WW + rx;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float rowacc = 0.f;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * w.WW + bb], wx[bb], rowacc);
        acc = fmaf(rowacc, wy[a], acc);
      }
    } else {
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
          rowacc = fmaf(val, wx[bb], rowacc);
        }
        acc = fmaf(rowacc, wy[a], acc);
      }
    }
    return acc;
  };
  for (int yl0 = tl.rs; yl0 < th; yl0 += tl.rstep * ADV_PF) {
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) {
      const int yl = yl0 + tl.rstep * d;          // wave-uniform
      const float cu = qu[d], cv = qv[d];
      {
        const int j = (ty0 + min(yl + tl.rstep * ADV_PF, th - 1)) * W + tl.cbase;
        qu[d] = srow(U + j)[tl.lx]; qv[d] = srow(V + j)[tl.lx];
      }
      if (yl < th) {
        const int y = ty0 + yl;
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, nullptr);
        int cell;
        const bool fast = tap_block_window<MODE>(ix, iy, Hpf, Wpf, Wf, wx0f, wy0f, WWf, WHf, tx, ty, cell);
        float acc;
        if (fast) {
          float wx[NT], wy[NT];
          Interp<MODE>::weights(tx, wx);
          Interp<MODE>::weights(ty, wy);
          const float* base = win + cell;
          acc = 0.f;
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            float rowacc = 0.f;
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[a * w.WW + bb], wx[bb], rowacc);
            acc = fmaf(rowacc, wy[a], acc);
          }
        } else {
          acc = slow_point(ix, iy);
        }
        if (tl.active) {
          srow(O + y * W + tl.cbase)[tl.lx] = acc;
        }
      }
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
  const double d = (double)x + 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
}

// mx = max |cotangent| of the tile, NaN if any cotangent is NaN (the reduction runs on the bit
// patterns of |g|: as unsigned integers they order like the floats, and every NaN sorts above +inf)
__device__ __forceinline__ void fixed_point_scale(float mx, float& scale, float& inv) {
  scale = 0.f; inv = 0.f;
  if (mx > 0.f && mx < INFINITY) {
    int e = 0;
    frexpf(mx, &e);                       // mx < 2^e
    e = e < -80 ? -80 : (e > 80 ? 80 : e);
    scale = ldexpf(1.0f, 40 - e);
    inv = ldexpf(1.0f, e - 40);
  } else if (!(mx < INFINITY)) {
    inv = NAN;                            // Inf or NaN cotangent: the field gradient is NaN, like float adds would give
  }
}

__device__ __forceinline__ unsigned abs_bits(float g) { return __float_as_uint(g) & 0x7fffffffu; }

__device__ __forceinline__ void departure_backward(const DepState& st, float sa, float ca, float gix,
                                                   float giy, const AdvGeom& g, float& gu, float& gv) {
  const float glam_c = gix * g.cx, gphi_c = giy * g.cy;
  const float sc = __builtin_amdgcn_fmed3f(st.s, -CLAMP_HI, CLAMP_HI);
  // d asin(s) / ds = 1 / sqrt(1 - s^2) = 1 / cos(lat_d), and 1 - s^2 = n^2 + d^2 identically ((s, n, d) is a unit
  // vector): the sum of squares has no cancellation, whereas 1 - s^2 formed from the rounded s loses
  // log2(1 / cos^2(lat_d)) bits next to the poles (at 89.3 degrees: relative error 8e-4 in fp32) - the reference's
  // fp32 autograd carries that error, the fp64 evaluation does not.
  // v_rsq / v_rcp (1 ulp) with one Newton step on the reciprocal: gradient error ~1e-7 relative
  const float den = fmaf(st.n, st.n, st.d * st.d);
#ifdef ADV_GS_FROM_S     // (diagnostic A/B builds only: the reference's form)
  const float gs = (sc == st.s) ? gphi_c * __builtin_amdgcn_rsqf(fmaf(-sc, sc, 1.0f)) : 0.f;
#else
  const float gs = (sc == st.s) ? gphi_c * __builtin_amdgcn_rsqf(den) : 0.f;
#endif
  float rden = __builtin_amdgcn_rcpf(den);
  rden = fmaf(fmaf(-den, rden, 1.0f), rden, rden);
  const float gl = glam_c * rden;
  const float gn = gl * st.d;
  const float gd = -gl * st.n;
  // d s / d phi' = cp ca - sp cl sa,   d n / d phi' = -sp sl,   d d / d phi' = -sp cl ca - cp sa
  const float spcl = st.sp * st.cl, cpsl = st.cp * st.sl;
  const float gphi = fmaf(gs, fmaf(st.cp, ca, -(spcl * sa)),
                          fmaf(gn, -(st.sp * st.sl), gd * -fmaf(spcl, ca, st.cp * sa)));
  // d s / d lam' = -cp sl sa,           d n / d lam' = cp cl,    d d / d lam' = -cp sl ca
  const float glam = fmaf(gs, -(cpsl * sa), fmaf(gn, st.cp * st.cl, gd * -(cpsl * ca)));
  gu = g.ndt * glam;
  gv = g.ndt * gphi;
}

// round-to-nearest-even integer of the product a*b in one double FMA (a, b widened once per row /
// column of the tap block instead of a conversion per tap): a*b is exact in double, the sum with
// 1.5*2^52 rounds it to an integer in the low mantissa bits
__device__ __forceinline__ unsigned long long fixed_from_product(double a, double b) {
  const double d = fma(a, b, 6755399441055744.0);
  return (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll);
}

// taps of one arrival point against a window: scatter g w_y w_x into the fixed-point accumulators,
// gather the field for the coordinate gradients.  base = index of tap (0,0) in the window.
template <int MODE>
__device__ __forceinline__ void scatter_gather(unsigned long long* acc, const float* win, int base, int WW,
                                               const float* wx, const float* wy, const float* dwx,
                                               const float* dwy, float gs_, float& gix, float& giy) {
  constexpr int NT = Interp<MODE>::NT;
  gix = 0.f; giy = 0.f;
  double wxd[NT];
#pragma unroll
  for (int bb = 0; bb < NT; ++bb) wxd[bb] = (double)wx[bb];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float sxv = 0.f, sdx = 0.f;
    const double gwy = (double)(gs_ * wy[a]);
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) {
      const int cell = base + a * WW + bb;
      const float val = win[cell];
      atomicAdd(&acc[cell], fixed_from_product(gwy, wxd[bb]));
      sxv = fmaf(val, wx[bb], sxv);
      sdx = fmaf(val, dwx[bb], sdx);
    }
    gix = fmaf(wy[a], sdx, gix);
    giy = fmaf(dwy[a], sxv, giy);
  }
}

// workgroup maximum of the |cotangent| bit patterns; contains a barrier
__device__ __forceinline__ float reduce_gmax(unsigned gmaxb, float* misc, int nwaves) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmaxb = max(gmaxb, (unsigned)__shfl_xor((int)gmaxb, o, 64));
  if ((threadIdx.x & 63) == 0) misc[2 + (threadIdx.x >> 6)] = __uint_as_float(gmaxb);
  __syncthreads();
  unsigned m = 0;
  for (int q = 0; q < nwaves; ++q) m = max(m, __float_as_uint(misc[2 + q]));
  return __uint_as_float(m);
}

// W == 64, separable grid: one workgroup per plane, wave w owns rows w, w+4, ...
// (five workgroups per CU: 29 KB of LDS each; the second launch-bound keeps the registers at 96)
#ifndef ADV_BWD_WAVES
#define ADV_BWD_WAVES 5
#endif
template <int MODE>
__global__ void __launch_bounds__(256, ADV_BWD_WAVES)
sl_advect_bwd_row64(const float* __restrict__ gout, const float* __restrict__ field,
                    const float* __restrict__ u, const float* __restrict__ v,
                    float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                    const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                    const float* __restrict__ lat_cells, const float* __restrict__ lon, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                    int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, int vec4) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0, W = 64;
  const int H = g.H, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;

  const int wn = Hp * Wp, wn2 = (wn + 1) & ~1;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);  // [wn] fixed-point sums
  float* win = smem + 2 * wn2;                                             // [wn]  F~ window
  float* misc = win + wn2;   // [0..1] pole means of gout, [2..6) per-wave max |cotangent|

  // first rows' operands while the window is staged
  float qu[ADV_PF], qv[ADV_PF], qg[ADV_PF];
#pragma unroll
  for (int d = 0; d < ADV_PF; ++d) {
    const int j = min(wave + 4 * d, H - 1) * W;
    qu[d] = srow(U + j)[lane]; qv[d] = srow(V + j)[lane]; qg[d] = srow(GO + j)[lane];
  }
  const float lonc = lon_cells(lon[lane], g);
  if (vec4) stage_plane_vec4(win, F, H, W, p);
  else {
    Window w{0, 0, Hp, Wp};
    stage_window(win, F, w, H, W, p, false, 0.f, 0.f, 256);
  }
  for (int i = tid; i < wn; i += 256) acc[i] = 0ull;
  unsigned gmaxb = 0;
  for (int y = wave; y < H; y += 4) {
    const float gval = srow(GO + y * W)[lane];
    gmaxb = max(gmaxb, abs_bits(gval));
    if (y == 0 || y == H - 1) {   // adjoint of the final pole mean: the cotangent of a pole row is its row mean
      const float m = wave_sum_dpp(gval) * (1.0f / 64.0f);
      if (lane == 0) misc[y == 0 ? 0 : 1] = m;
    }
  }
  const float mxall = reduce_gmax(gmaxb, misc, 4);    // its barrier also closes the staging
  pole_rows_to_mean_lds(win, H, W, p, Wp);
  __syncthreads();
  const float gm0 = misc[0], gm1 = misc[1];
  float scale, inv_scale;   // every thread derives the same power-of-two scale
  fixed_point_scale(mxall, scale, inv_scale);

  const float Hpf = (float)Hp, Wpf = (float)Wp;
  for (int y0 = wave; y0 < H; y0 += 4 * ADV_PF) {
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) {
      const int y = y0 + 4 * d;                 // wave-uniform
      const float cu = qu[d], cv = qv[d], cgo = qg[d];
      {
        const int j = min(y + 4 * ADV_PF, H - 1) * W;
        qu[d] = srow(U + j)[lane]; qv[d] = srow(V + j)[lane]; qg[d] = srow(GO + j)[lane];
      }
      if (y < H) {
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty, wx[NT], wy[NT], dwx[NT], dwy[NT];
        DepState st;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, &st);
        int cell;
        const bool edge = tap_block_whole<MODE>(ix, iy, Hpf, Wpf, tx, ty, cell);
        int base = cell + OFF0 * (Wp + 1);
        if (__any(edge)) {
          int bx, by, sx, sy;
          tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
          Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
          Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
          shift_weights<NT>(wx, sx); shift_weights<NT>(dwx, sx);
          shift_weights<NT>(wy, sy); shift_weights<NT>(dwy, sy);
          base = by * Wp + bx;
        } else {
          Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
          Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
        }
        const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : cgo);
        float gix, giy;
        scatter_gather<MODE>(acc, win, base, Wp, wx, wy, dwx, dwy, gval * scale, gix, giy);
        float guv, gvv;
        departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
        srow(GU + y * W)[lane] = guv;
        srow(GV + y * W)[lane] = gvv;
      }
    }
  }
  __syncthreads();
  // fold the halo back: every source cell sums its aliases (adjoint of the a1 map: its own cell, the
  // lon-wrap copies of the p edge columns, and for rows next to a pole the mirrored row shifted by
  // W/2), then the adjoint of the first pole mean (rows 0, H-1 <- their mean)
  const double inv = (double)inv_scale;
  const bool lo_edge = lane < p, hi_edge = lane >= W - p;
  const unsigned xm = lane ^ 32u;                     // (x + W/2) mod W
  const bool mlo = xm < p, mhi = xm >= W - p;
  for (int y = wave; y < H; y += 4) {
    int mr = -1;                                 // padded row of the over-the-pole alias (wave-uniform)
    if (y >= 1 && y <= p) mr = p - y;
    else if (y >= H - 1 - p && y <= H - 2) mr = 2 * (H - 1) - y + p;
    const unsigned long long* row = acc + (y + p) * Wp + p;
    long long s = (long long)row[lane];
    if (lo_edge) s += (long long)row[lane + W];
    if (hi_edge) s += (long long)row[lane - W];
    if (mr >= 0) {
      const unsigned long long* mrow = acc + mr * Wp + p;
      s += (long long)mrow[xm];
      if (mlo) s += (long long)mrow[xm + W];
      if (mhi) s += (long long)mrow[xm - W];
    }
    float val = (float)((double)s * inv);
    if (y == 0 || y == H - 1) val = wave_sum_dpp(val) * (1.0f / 64.0f);
    srow(GF + y * W)[lane] = val;
  }
}

template <int MODE, bool WHOLE, int NTH, bool DET = false>   // DET: integer global accumulators (deterministic tiled mode)
__global__ void __launch_bounds__(NTH)
sl_advect_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ field,
                     const float* __restrict__ u, const float* __restrict__ v,
                     float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                     const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lon, const float* __restrict__ fmeans,
                     const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                     int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, int halo, int tiles_x, int tiles,
                     int vec4, unsigned long long* __restrict__ gacc, const unsigned* __restrict__ pmax) {
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
  unsigned gmaxb = WHOLE ? 0u : max(abs_bits(gm0), abs_bits(gm1));
  {
    TileIter itg(tid, tw, NTH);
    for (int i = tid; i < npts; i += NTH, itg.next())
      gmaxb = max(gmaxb, abs_bits(GO[(ty0 + itg.yl) * W + tx0 + itg.xl]));
  }
  float mxall = reduce_gmax(gmaxb, misc, NTH / 64);
  // deterministic tiled mode: integer global accumulators need ONE scale per plane (its max |cotangent|, a pre-pass)
  unsigned long long* GA = nullptr;
  if constexpr (DET) { GA = gacc + (int64_t)plane * P; mxall = __uint_as_float(pmax[plane]); }
  if (WHOLE) {
    if (wave < 2) {
      float* row = win + (wave == 0 ? p : H - 1 + p) * Wp;
      const float m = wave_row_mean(row + p, W);
      for (int x = tid & 63; x < Wp; x += 64) row[x] = m;
    } else if (wave < 4) {
      // adjoint of the final pole mean: the cotangent of a pole row is its own row mean
      const float* row = GO + (wave == 2 ? 0 : (int64_t)(H - 1) * W);
      const float m = wave_row_mean(row, W);
      if ((tid & 63) == 0) misc[wave - 2] = m;
    }
    __syncthreads();
    gm0 = misc[0]; gm1 = misc[1];
  }
  float scale, inv_scale;   // every thread derives the same power-of-two scale
  fixed_point_scale(mxall, scale, inv_scale);

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
    departure(cu, cv, sa, ca, lon_cells(clo, g), g, ix, iy, &st);
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
    int ry = by - w.wy0, rx = bx - w.wx0;
    bool inwin = true;
    if (!WHOLE) {
      if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
      inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    }
    float gix = 0.f, giy = 0.f;
    if (inwin) {
      scatter_gather<MODE>(acc, win, ry * w.WW + rx, w.WW, wx, wy, dwx, dwy, gval * scale, gix, giy);
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
          if constexpr (DET) atomicAdd(&GA[(int64_t)r * W + c], fixed_from_product((double)(gval * scale * wy[a]), (double)wx[bb]));
          else atomicAdd(&GF[(int64_t)r * W + c], gval * wy[a] * wx[bb]);
          sxv = fmaf(val, wx[bb], sxv);
          sdx = fmaf(val, dwx[bb], sdx);
        }
        gix = fmaf(wy[a], sdx, gix);
        giy = fmaf(dwy[a], sxv, giy);
      }
    }
    float guv, gvv;
    departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
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
      win[i] = (float)((double)s * inv);   // the float plane reuses the window storage
    }
    __syncthreads();
    if (wave < 2) {
      float* row = win + (wave == 0 ? 0 : (H - 1) * W);
      const float m = wave_row_mean(row, W);
      for (int x = tid & 63; x < W; x += 64) row[x] = m;
    }
    __syncthreads();
    for (int i = tid; i < P; i += NTH) {
      GF[i] = win[i];
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
      if constexpr (DET) atomicAdd(&GA[(int64_t)sr * W + sc], (unsigned long long)s);   // integer: order-independent
      else atomicAdd(&GF[(int64_t)sr * W + sc], (float)((double)s * inv));
    }
  }
}


// Backward of the tiled schedule on a separable grid, one wave per 64 columns of a tile row (see
// sl_advect_fwd_tilerow); window accumulators, flush and the global-atomic fallback as in the generic
// tiled kernel below.
template <int MODE, bool DET = false>
__global__ void __launch_bounds__(TILED_THREADS_BWD, 4)   // four waves per SIMD: <= 128 VGPRs
sl_advect_bwd_tilerow(const float* __restrict__ gout, const float* __restrict__ field,
                      const float* __restrict__ u, const float* __restrict__ v,
                      float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                      const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                      const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans,
                      const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                      int64_t uv_bs, int64_t gf_bs, int64_t guv_bs, int halo, int tiles_x, int tiles,
                      unsigned long long* __restrict__ gacc, const unsigned* __restrict__ pmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = Interp<MODE>::NT, NTH = TILED_THREADS_BWD;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x / tiles, tile = blockIdx.x - plane * tiles;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;
  const int ty0 = (tile / tiles_x) * TILE_H, tx0 = (tile % tiles_x) * TILE_W;
  const int th = min(TILE_H, H - ty0), tw = min(TILE_W, W - tx0);
  Window w{ty0 + p - halo, tx0 + p - halo, TILE_H + 2 * halo + NT, TILE_W + 2 * halo + NT};
  const int wn = w.WH * w.WW, wn2 = (wn + 1) & ~1;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);  // [wn] fixed-point sums
  float* win = smem + 2 * wn2;                                             // [wn]  F~ window
  float* misc = win + wn2;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const float gm0 = gmeans[2 * plane], gm1 = gmeans[2 * plane + 1];
  const TileRowLane tl(wave, NTH / 64, lane, tx0, tw, W);

  float qu[ADV_PF], qv[ADV_PF], qg[ADV_PF];
#pragma unroll
  for (int d = 0; d < ADV_PF; ++d) {
    const int j = (ty0 + min(tl.rs + tl.rstep * d, th - 1)) * W + tl.cbase;
    qu[d] = srow(U + j)[tl.lx]; qv[d] = srow(V + j)[tl.lx]; qg[d] = srow(GO + j)[tl.lx];
  }
  const float lonc = lon_cells(srow(lon + tl.cbase)[tl.lx], g);
  stage_window(win, F, w, H, W, p, true, m0, m1, NTH);
  for (int i = tid; i < wn; i += NTH) acc[i] = 0ull;
  unsigned gmaxb = max(abs_bits(gm0), abs_bits(gm1));
  for (int yl = tl.rs; yl < th; yl += tl.rstep)
    gmaxb = max(gmaxb, abs_bits(srow(GO + (ty0 + yl) * W + tl.cbase)[tl.lx]));
  float mxall = reduce_gmax(gmaxb, misc, NTH / 64);    // its barrier also closes the staging
  // deterministic mode: integer global accumulators, ONE scale per plane (its max |cotangent|, a pre-pass)
  unsigned long long* GA = nullptr;
  if constexpr (DET) { GA = gacc + (int64_t)plane * P; mxall = __uint_as_float(pmax[plane]); }
  float scale, inv_scale;
  fixed_point_scale(mxall, scale, inv_scale);

  const float Hpf = (float)Hp, Wpf = (float)Wp, Wf = (float)W, wx0f = (float)w.wx0, wy0f = (float)w.wy0,
              WWf = (float)w.WW, WHf = (float)w.WH;
  // general path of a point: any tap origin; taps against the window or, outside it, against global
  // memory through the index map (float atomics on the field gradient)
  auto slow_point = [&](float ix, float iy, float gval, float& gix, float& giy) {
    float tx, ty, wx[NT], wy[NT], dwx[NT], dwy[NT];
    int bx, by, sx, sy;
    tap_origin<MODE>(ix, iy, Hp, Wp, bx, by, sx, sy, tx, ty);
    Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
    Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
    shift_weights<NT>(wx, sx); shift_weights<NT>(dwx, sx);
    shift_weights<NT>(wy, sy); shift_weights<NT>(dwy, sy);
    int ry = by - w.wy0, rx = bx - w.wx0;
    if (rx < 0) rx += W; else if (rx > w.WW - NT) rx -= W;
    const bool inwin = ry >= 0 && ry <= w.WH - NT && rx >= 0 && rx <= w.WW - NT;
    if (inwin) {
      scatter_gather<MODE>(acc, win, ry * w.WW + rx, w.WW, wx, wy, dwx, dwy, gval * scale, gix, giy);
      return;
    }
    gix = 0.f; giy = 0.f;
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
        if constexpr (DET) atomicAdd(&GA[(int64_t)r * W + c], fixed_from_product((double)(gval * scale * wy[a]), (double)wx[bb]));
        else atomicAdd(&GF[(int64_t)r * W + c], gval * wy[a] * wx[bb]);
        sxv = fmaf(val, wx[bb], sxv);
        sdx = fmaf(val, dwx[bb], sdx);
      }
      gix = fmaf(wy[a], sdx, gix);
      giy = fmaf(dwy[a], sxv, giy);
    }
  };
  for (int yl0 = tl.rs; yl0 < th; yl0 += tl.rstep * ADV_PF) {
#pragma unroll
    for (int d = 0; d < ADV_PF; ++d) {
      const int yl = yl0 + tl.rstep * d;          // wave-uniform
      const float cu = qu[d], cv = qv[d], cgo = qg[d];
      {
        const int j = (ty0 + min(yl + tl.rstep * ADV_PF, th - 1)) * W + tl.cbase;
        qu[d] = srow(U + j)[tl.lx]; qv[d] = srow(V + j)[tl.lx]; qg[d] = srow(GO + j)[tl.lx];
      }
      if (yl < th) {
        const int y = ty0 + yl;
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty;
        DepState st;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, &st);
        int cell;
        const bool fast = tap_block_window<MODE>(ix, iy, Hpf, Wpf, Wf, wx0f, wy0f, WWf, WHf, tx, ty, cell);
        const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : cgo);
        float gix = 0.f, giy = 0.f;
        if (tl.active) {
          if (fast) {
            float wx[NT], wy[NT], dwx[NT], dwy[NT];
            Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
            Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
            scatter_gather<MODE>(acc, win, cell, w.WW, wx, wy, dwx, dwy, gval * scale, gix, giy);
          } else {
            slow_point(ix, iy, gval, gix, giy);
          }
          float guv, gvv;
          departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
          srow(GU + y * W + tl.cbase)[tl.lx] = guv;
          srow(GV + y * W + tl.cbase)[tl.lx] = gvv;
        }
      }
    }
  }
  __syncthreads();
  // flush the window once: one global float atomic per touched cell instead of 16 per point.  A thread
  // keeps its window column (longitude wrap and pole shift computed once) and walks down the rows.
  const double inv = (double)inv_scale;
  {
    const int cols = w.WW < NTH ? w.WW : NTH, rpp = NTH / cols;
    const int r0 = tid / cols, c0 = tid - r0 * cols;
    if (r0 < rpp) {
      for (int lc = c0; lc < w.WW; lc += cols) {
        int jj = (w.wx0 + lc - p) % W;
        if (jj < 0) jj += W;
        int jm = jj + (W >> 1);
        if (jm >= W) jm -= W;
        for (int lr = r0; lr < w.WH; lr += rpp) {
          const long long sv = (long long)acc[lr * w.WW + lc];
          const int r = w.wy0 + lr;
          if (sv == 0 || r < 0 || r >= Hp) continue;
          const int ii = r - p;
          const int sr = ii < 0 ? -ii : (ii >= H ? 2 * (H - 1) - ii : ii);
          const int sc = (ii < 0 || ii >= H) ? jm : jj;
          if constexpr (DET) atomicAdd(&GA[(int64_t)sr * W + sc], (unsigned long long)sv);   // integer: order-independent
          else atomicAdd(&GF[(int64_t)sr * W + sc], (float)((double)sv * inv));
        }
      }
    }
  }
}

// ======================================================================================
// strip schedule (round 4): large planes on a separable grid (configs[3], configs[4])
// ======================================================================================
// A workgroup owns a STRIP of 128 arrival columns of one plane and walks down ALL its rows, eight per step.  The window
// of the padded plane it samples from is a RING of R padded rows x (128 + 2 hx + NT) columns in LDS: while the eight
// arrival rows of step s are computed against rows [lo(s), lo(s) + R - 8), the eight rows that enter the window at step
// s + 1 are written into the ring slots of the eight rows that left it at step s - 1 (their global loads were issued a
// step earlier and sit in registers), so one barrier per step separates readers and writers and staging overlaps with
// the arithmetic.  What this buys over the 64 x 128 / 16 x 128 tiles of rounds 1-3:
//   * a window cell is staged once per strip, not once per tile row band: 1.19 (hx = 10) to 1.5 (hx = 32) cells per
//     arrival point instead of 1.5-3.0 - the latitude halo costs LDS rows, not HBM bytes - and the halos can be
//     asymmetric: displacements are a few rows in latitude but tens of columns in longitude next to the poles
//     (tools/adv_disp_stats.py: in the default model at 721 x 1440 |dy| stays below 22 rows for 99 % of the points
//     while 30 % of them move more than 16 columns);
//   * backward: the accumulator rows leave the ring complete - one float atomic per window cell of the strip,
//     1.19 per point instead of 3.0 - and the power-of-two scale of the 64-bit fixed-point sums follows the running
//     maximum of the cotangent (a rescale of the ring when a step raises it) instead of a pre-pass over the cotangent;
//   * taps that leave the window go to L2 with the row / column maps of the tap block evaluated once per point
//     (4 + 4 index computations instead of 16).
constexpr int STRIP_W = 128, STRIP_ROWS = 8, STRIP_THREADS = 512, STRIP_CH = 4;   // CH: 64-column chunks of a window row
constexpr int STRIP_MAX_WS = 64 * STRIP_CH;
constexpr int STRIP_LDS_MAX = 160 * 1024 - 1024;   // dynamic LDS a strip kernel may ask for (it also holds a static counter word)

// lane's source columns of the window columns lane, lane + 64, ...: image column in the low half, the column of the
// over-the-pole rows (shifted by W / 2) in the high half
struct StripCols { unsigned pk[STRIP_CH]; };
__device__ __forceinline__ StripCols strip_cols(int wx0, int WS, int W, int p, unsigned lane, int cfirst = 0) {
  StripCols c;
#pragma unroll
  for (int i = 0; i < STRIP_CH; ++i) {
    const int lc = min((int)lane + 64 * (cfirst + i), WS - 1);
    int jj = (wx0 + lc - p) % W;
    if (jj < 0) jj += W;
    int jm = jj + (W >> 1);
    if (jm >= W) jm -= W;
    c.pk[i] = (unsigned)jj | ((unsigned)jm << 16);
  }
  return c;
}

// source row of padded row pr (wave-uniform): valid = inside the padded plane, mir = beyond a pole
struct StripRow { int sr; bool valid, mir; };
__device__ __forceinline__ StripRow strip_row(int pr, int H, int p) {
  const int Hp = H + 2 * p;
  StripRow r;
  r.valid = pr >= 0 && pr < Hp;
  const int ii = min(max(pr, 0), Hp - 1) - p;
  r.mir = ii < 0 || ii >= H;
  r.sr = ii < 0 ? -ii : (ii >= H ? 2 * (H - 1) - ii : ii);
  return r;
}

__device__ __forceinline__ void strip_load_row(const float* __restrict__ F, int pr, int H, int W, int p, const StripCols& cs,
                                               int nch, float (&pre)[STRIP_CH]) {
  const StripRow r = strip_row(pr, H, p);
  const global_ptr<const float> row = srow(F + (int64_t)r.sr * W);
#pragma unroll
  for (int i = 0; i < STRIP_CH; ++i)
    if (i < nch) pre[i] = row[r.mir ? (cs.pk[i] >> 16) : (cs.pk[i] & 0xffffu)];
}

__device__ __forceinline__ void strip_store_row(float* __restrict__ ring, int pr, int RMASK, int WS, int H, int p, float m0,
                                                float m1, unsigned lane, int nch, const float (&pre)[STRIP_CH],
                                                int cfirst = 0) {
  const StripRow r = strip_row(pr, H, p);
  float* dst = ring + (pr & RMASK) * WS;
#pragma unroll
  for (int i = 0; i < STRIP_CH; ++i)
    if (i < nch) {
      float v = pre[i];
      if (r.sr == 0) v = m0;                 // rows 0 and H-1 enter as their longitudinal means (advection.py:100-114)
      if (r.sr == H - 1) v = m1;
      if (!r.valid) v = 0.f;
      const int lc = (int)lane + 64 * (cfirst + i);
      if (lc < WS) dst[lc] = v;
    }
}

// lanes of a wave inside a 128-column strip: waves 2r, 2r+1 own the two 64-column halves of a row
struct StripLane {
  int cbase; unsigned lx; bool active;
  __device__ __forceinline__ StripLane(int half, unsigned lane, int x0, int tw, int W) {
    const int first = half * 64;
    active = first + (int)lane < tw;
    cbase = min(x0 + first, W - 1);
    lx = min(lane, (unsigned)max(tw - 1 - first, 0));
  }
};

// column maps of a tap block at clamped origin column bx (padded coordinates): col[b] / colm[b] = source column of a
// plain / over-the-pole row
template <int NT>
struct TapCols { int col[NT], colm[NT]; };
template <int NT>
__device__ __forceinline__ void tap_cols(int bx, int W, int p, TapCols<NT>& m) {
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    int j = bx + b - p;
    j += j < 0 ? W : 0;
    j -= j >= W ? W : 0;
    int jm = j + (W >> 1);
    jm -= jm >= W ? W : 0;
    m.col[b] = j; m.colm[b] = jm;
  }
}
// source row of padded tap row pr: returns the row offset sr * W; mir = beyond a pole, pole = 1 / 2 for rows 0 / H-1
__device__ __forceinline__ int tap_row(int pr, int H, int W, int p, bool& mir, int& pole) {
  const int ii = pr - p;
  mir = ii < 0 || ii >= H;
  const int sr = ii < 0 ? -ii : (ii >= H ? 2 * (H - 1) - ii : ii);
  pole = sr == 0 ? 1 : (sr == H - 1 ? 2 : 0);
  return sr * W;
}

// value of a point whose tap block leaves the window: all taps from global memory (L2) through the a1 map, two tap
// rows in flight at a time (the registers of this path set the occupancy of the whole kernel)
template <int MODE>
__device__ __forceinline__ float gather_global(const float* __restrict__ F, float ix, float iy, int H, int W, int p,
                                               float m0, float m1) {
  constexpr int NT = Interp<MODE>::NT;
  float tx, ty, wx[NT], wy[NT];
  int bx, by, sx, sy;
  tap_origin<MODE>(ix, iy, H + 2 * p, W + 2 * p, bx, by, sx, sy, tx, ty);
  Interp<MODE>::weights(tx, wx);
  Interp<MODE>::weights(ty, wy);
  shift_weights<NT>(wx, sx);
  shift_weights<NT>(wy, sy);
  TapCols<NT> m;
  tap_cols<NT>(bx, W, p, m);
  float acc = 0.f;
#pragma unroll
  for (int a0 = 0; a0 < NT; a0 += 2) {
    float val[2][NT];
    int pole[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bool mir;
      const int ro = tap_row(by + a0 + q, H, W, p, mir, pole[q]);
#pragma unroll
      for (int b = 0; b < NT; ++b) val[q][b] = F[ro + (mir ? m.colm[b] : m.col[b])];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float rowacc = 0.f;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const float t = pole[q] == 1 ? m0 : (pole[q] == 2 ? m1 : val[q][b]);
        rowacc = fmaf(t, wx[b], rowacc);
      }
      acc = fmaf(rowacc, wy[a0 + q], acc);
    }
    asm volatile("" ::: "memory");      // keeps the next pair's loads behind this pair's use
  }
  return acc;
}

// tap block of a point against the ring window: fraction, plane clamp, window-relative column (the window may
// straddle the date line), padded row of tap row 0.  Returns true when the NT x NT block lies inside the resident
// rows [rlo, rhi] x the window columns and the plane clamp did not move it.  All values are integers below 2^24 held
// in floats; r0 / c0 = padded row / window column of tap (0, 0).
template <int MODE>
__device__ __forceinline__ bool tap_block_ring(float ix, float iy, float Hpf, float Wpf, float Wf, float wx0f, float WSf,
                                               float rlof, float rhif, float& tx, float& ty, int& r0, int& c0) {
  constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  tx = __builtin_amdgcn_fractf(ix);
  ty = __builtin_amdgcn_fractf(iy);
  const float x0f = ix - tx, y0f = iy - ty;
  const float xc = __builtin_amdgcn_fmed3f(x0f, (float)(-OFF0), Wpf - (float)(NT + OFF0));
  const float yc = __builtin_amdgcn_fmed3f(y0f, (float)(-OFF0), Hpf - (float)(NT + OFF0));
  float rx = (xc + (float)OFF0) - wx0f;
  const float ry = yc + (float)OFF0;
  rx = rx < 0.f ? rx + Wf : (rx > WSf - (float)NT ? rx - Wf : rx);
  const float rxc = __builtin_amdgcn_fmed3f(rx, 0.f, WSf - (float)NT);
  const float ryc = __builtin_amdgcn_fmed3f(ry, rlof, rhif - (float)(NT - 1));
  r0 = (int)ryc;
  c0 = (int)rxc;
  return xc == x0f && yc == y0f && rxc == rx && ryc == ry;
}

// append this lane's point to the strip's list of deferred points: one LDS atomic per wave that holds any
// (forward: the sample coordinates ride along in two more lists of the same length behind the first, cap entries each)
__device__ __forceinline__ void strip_defer(bool defer, unsigned idx, unsigned* __restrict__ q, unsigned* qcount,
                                            unsigned lane, size_t cap = 0, float ix = 0.f, float iy = 0.f) {
  const unsigned long long m = __ballot(defer);
  if (m == 0ull) return;                                  // wave-uniform
  const int leader = __ffsll((long long)m) - 1;
  unsigned base = 0;
  if ((int)lane == leader) base = atomicAdd(qcount, (unsigned)__popcll(m));
  base = __shfl((int)base, leader, 64);
  if (defer) {
    const unsigned slot = base + __popcll(m & ((1ull << lane) - 1ull));
    q[slot] = idx;
    if (cap) {
      reinterpret_cast<float*>(q)[cap + slot] = ix;
      reinterpret_cast<float*>(q)[2 * cap + slot] = iy;
    }
  }
}

// geometry of the ring: rows resident while the arrival rows [8 s, 8 s + 8) are computed
template <int MODE, int R>
struct StripRing {
  static constexpr int NT = Interp<MODE>::NT, OFF0 = Interp<MODE>::OFF0;
  static constexpr int RW = R - STRIP_ROWS;                      // resident rows
  static constexpr int HYL = (RW - (STRIP_ROWS - 1 + NT)) / 2;     // latitude halo below (rows); above: the rest
  static __device__ __forceinline__ int lo(int s, int p) { return STRIP_ROWS * s + p + OFF0 - HYL; }
};

// waves per SIMD the strip kernels are compiled for: the forward's 19 KB ring of the 128-row grids admits four
// 8-wave workgroups per CU (64 registers, no scratch), its 50 KB ring of the large grids three (74 registers)
#ifndef STRIP_FWD_WAVES
#define STRIP_FWD_WAVES (R == 32 ? 8 : 6)
#endif
#ifndef STRIP_BWD_WAVES
#define STRIP_BWD_WAVES 4
#endif
#ifndef STRIP_FWD_INLINE_FIXUP     // 1: a strip's workgroup processes its own deferred points; 0: a second launch does
#define STRIP_FWD_INLINE_FIXUP 1
#endif
template <int MODE, int R>
__global__ void __launch_bounds__(STRIP_THREADS, STRIP_FWD_WAVES)
sl_advect_fwd_strip(const float* __restrict__ field, const float* __restrict__ u, const float* __restrict__ v,
                    float* __restrict__ out, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                    const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans,
                    int K, AdvGeom g, int64_t f_bs, int64_t uv_bs, int64_t o_bs, int hx, int strips,
                    unsigned* __restrict__ queue, unsigned* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ unsigned qcount;
  using Ring = StripRing<MODE, R>;
  constexpr int NT = Interp<MODE>::NT, RW = Ring::RW, RMASK = R - 1;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x / strips, strip = blockIdx.x - plane * strips;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;
  const int x0 = strip * STRIP_W, tw = min(STRIP_W, W - x0);
  const int WS = STRIP_W + 2 * hx + NT, wx0 = x0 + p - hx, nch = (WS + 63) >> 6;
  float* ring = smem;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const StripLane tl(wave & 1, lane, x0, tw, W);
  const int rw = wave >> 1;                          // this wave's arrival rows of a step: rw and rw + 4
  const int nsteps = (H + STRIP_ROWS - 1) / STRIP_ROWS;
  const StripCols cs = strip_cols(wx0, WS, W, p, lane);
  const size_t qcap = (size_t)H * STRIP_W;
  unsigned* const q = queue + (size_t)blockIdx.x * 3 * qcap;    // this strip's lists of deferred points: index, ix, iy
  if (tid == 0) qcount = 0u;

  float qu[2], qv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = min(rw + 4 * j, H - 1) * W + tl.cbase;
    qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx];
  }
  const float lonc = lon_cells(srow(lon + tl.cbase)[tl.lx], g);

  // initial window: rows [lo(0), lo(0) + RW), two rounds of eight rows in flight
  float pre[STRIP_CH], pre2[STRIP_CH];
  const int lo0 = Ring::lo(0, p);
  for (int r8 = 0; r8 < RW; r8 += 2 * STRIP_ROWS) {
    strip_load_row(F, lo0 + r8 + wave, H, W, p, cs, nch, pre);
    if (r8 + STRIP_ROWS < RW) strip_load_row(F, lo0 + r8 + STRIP_ROWS + wave, H, W, p, cs, nch, pre2);
    strip_store_row(ring, lo0 + r8 + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre);
    if (r8 + STRIP_ROWS < RW) strip_store_row(ring, lo0 + r8 + STRIP_ROWS + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre2);
  }
  strip_load_row(F, lo0 + RW + wave, H, W, p, cs, nch, pre);      // the rows that enter at step 1
  __syncthreads();

  const float Hpf = (float)Hp, Wpf = (float)Wp, Wf = (float)W, wx0f = (float)wx0, WSf = (float)WS;
  for (int s = 0; s < nsteps; ++s) {
    const int lo = Ring::lo(s, p);
    // rows [lo + RW, lo + R): ring slots nobody reads during this step
    strip_store_row(ring, lo + RW + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre);
    if (s + 1 < nsteps) strip_load_row(F, lo + R + wave, H, W, p, cs, nch, pre);
    const float rlof = (float)lo, rhif = (float)(lo + RW - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = STRIP_ROWS * s + rw + 4 * j;          // wave-uniform
      const float cu = qu[j], cv = qv[j];
      {
        const int idx = min(y + STRIP_ROWS, H - 1) * W + tl.cbase;
        qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx];
      }
      if (y < H) {
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, nullptr);
        int r0, c0;
        const bool fast = tap_block_ring<MODE>(ix, iy, Hpf, Wpf, Wf, wx0f, WSf, rlof, rhif, tx, ty, r0, c0);
        // (a tap block outside the window reads a clamped, valid address; its value is dropped)
        float wx[NT], wy[NT];
        Interp<MODE>::weights(tx, wx);
        Interp<MODE>::weights(ty, wy);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          const float* base = ring + ((r0 + a) & RMASK) * WS + c0;
          float rowacc = 0.f;
#pragma unroll
          for (int bb = 0; bb < NT; ++bb) rowacc = fmaf(base[bb], wx[bb], rowacc);
          acc = fmaf(rowacc, wy[a], acc);
        }
        if (tl.active && fast) srow(O + y * W + tl.cbase)[tl.lx] = acc;
        strip_defer(!fast && tl.active, (unsigned)(y * W + tl.cbase) + tl.lx, q, &qcount, lane, qcap, ix, iy);
      }
    }
    // raw barrier: __syncthreads() would also wait for this step's stores and the prefetches (s_waitcnt vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if constexpr (STRIP_FWD_INLINE_FIXUP && R == 64) {
  // (large grids only: with the 32-row ring the extra code costs the kernel its 64-register fit, and the few deferred
  //  points of the 128-row grids are cheap in the second launch: 0.93 vs 0.96 ms per launch in the model)
  // The deferred points of this strip, densely, by the workgroup that listed them: while the other workgroups of the CU
  // are still in their row loops the scattered 4-byte loads of this phase overlap with their arithmetic, and the rows
  // it touches were streamed through the caches moments ago.  (The lists were written by this workgroup's own waves:
  // __syncthreads() drains the stores; no line of a list has been read before, so no stale L1 copy exists.)
  __syncthreads();
  {
    const unsigned n = qcount;
    const float* qx = reinterpret_cast<const float*>(q) + qcap;
    const float* qy = qx + qcap;
    for (unsigned i = tid; i < n; i += STRIP_THREADS)
      O[q[i]] = gather_global<MODE>(F, qx[i], qy[i], H, W, p, m0, m1);
  }
  if (tid == 0) counts[blockIdx.x] = 0u;
  } else {
    if (tid == 0) counts[blockIdx.x] = qcount;
  }
}

// Deferred points of the strips (tap block outside the ring window: the polar rows, and whatever moved further than the
// halos): every lane takes one point - the divergent slow path of rounds 1-3 ran the sixteen global loads of a point
// for every WAVE that held one; here the points of a strip are processed densely, in a launch of their own.
template <int MODE>
__global__ void __launch_bounds__(256)
sl_advect_fwd_strip_fixup(const float* __restrict__ field, const float* __restrict__ u, const float* __restrict__ v,
                          float* __restrict__ out, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                          const float* __restrict__ lon, const float* __restrict__ fmeans, int K, AdvGeom g, int64_t f_bs,
                          int64_t uv_bs, int64_t o_bs, int strips, const unsigned* __restrict__ queue,
                          const unsigned* __restrict__ counts) {
  const int H = g.H, W = g.W, p = g.p, P = H * W;
  const unsigned n = counts[blockIdx.x];
  if (n == 0) return;
  const int plane = blockIdx.x / strips;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;
  const size_t qcap = (size_t)H * STRIP_W;
  const unsigned* q = queue + (size_t)blockIdx.x * 3 * qcap;
  const float* qx = reinterpret_cast<const float*>(q) + qcap;
  const float* qy = qx + qcap;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  for (unsigned i = threadIdx.x; i < n; i += 256)
    O[q[i]] = gather_global<MODE>(F, qx[i], qy[i], H, W, p, m0, m1);
}

#ifdef PARADIS_DEV_KNOBS
// Diagnostic (round 6, verdict r5 item 5; development library only, PARADIS_ADVECT_DIRECT=4|8): the forward gather with NO LDS window - every tap a
// global load served by L2.  All workgroups of a plane run on ONE XCD (blockIdx % 8 = XCD; a 721 x 1440 plane is 4.15 MB,
// that XCD's L2 holds the band of rows in flight), a wave takes 64 points of one latitude row (scalar sa / ca as in the
// strips), a workgroup DROWS consecutive rows so that its waves share tap rows through the CU's L1.  Same departure
// point and tap arithmetic as the strips' deferred points (gather_global).  A/B (profiles/r06_advect_direct.txt, tools/advect_direct_ab.py):
// 2-3x slower than the strips with bicubic taps, 1.5x with bilinear, outputs identical.
template <int MODE, int DROWS>
__global__ void __launch_bounds__(64 * DROWS)
sl_advect_fwd_direct(const float* __restrict__ field, const float* __restrict__ u, const float* __restrict__ v,
                     float* __restrict__ out, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans,
                     int K, AdvGeom g, int64_t f_bs, int64_t uv_bs, int64_t o_bs, int planes, int rgroups, int cblocks) {
  const int H = g.H, W = g.W, p = g.p, P = H * W;
  const int per_plane = rgroups * cblocks;
  const unsigned j = blockIdx.x >> 3;
  const int plane = (int)(j / per_plane) * 8 + (int)(blockIdx.x & 7);
  if (plane >= planes) return;
  const int t = (int)(j % per_plane), rg = t / cblocks, cb = t - rg * cblocks;
  const unsigned lane = threadIdx.x & 63;
  const int y = rg * DROWS + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (y >= H) return;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;
  const int x = cb * 64 + (int)lane;
  const bool active = x < W;
  const unsigned xc = (unsigned)min(x, W - 1);
  const float cu = srow(U + y * W)[xc], cv = srow(V + y * W)[xc];
  const float lonc = lon_cells(srow(lon)[xc], g);
  const float sa = sin_lat[y * W], ca = cos_lat[y * W];
  float ix, iy;
  departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, nullptr);
  const float r = gather_global<MODE>(F, ix, iy, H, W, p, fmeans[2 * plane], fmeans[2 * plane + 1]);
  if (active) srow(O + y * W)[xc] = r;
}
#endif

// field gradient and coordinate gradients of a point whose tap block leaves the window: global atomics on the field
// gradient (float, or the integer plane of the deterministic mode), taps from global memory, one tap row at a time
template <int MODE, bool DET>
__device__ __forceinline__ void scatter_global(const float* __restrict__ F, float* __restrict__ GF,
                                               unsigned long long* __restrict__ GA, float ix, float iy, int H, int W,
                                               int p, float m0, float m1, float gval, float scale, float& gix, float& giy) {
  constexpr int NT = Interp<MODE>::NT;
  float tx, ty, wx[NT], wy[NT], dwx[NT], dwy[NT];
  int bx, by, sx, sy;
  tap_origin<MODE>(ix, iy, H + 2 * p, W + 2 * p, bx, by, sx, sy, tx, ty);
  Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
  Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
  shift_weights<NT>(wx, sx); shift_weights<NT>(dwx, sx);
  shift_weights<NT>(wy, sy); shift_weights<NT>(dwy, sy);
  TapCols<NT> m;
  tap_cols<NT>(bx, W, p, m);
  gix = 0.f; giy = 0.f;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    bool mir;
    int pole;
    const int ro = tap_row(by + a, H, W, p, mir, pole);
    float val[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) val[b] = F[ro + (mir ? m.colm[b] : m.col[b])];
    float sxv = 0.f, sdx = 0.f;
    const float gwy = gval * wy[a];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int cell = ro + (mir ? m.colm[b] : m.col[b]);
      const float t = pole == 1 ? m0 : (pole == 2 ? m1 : val[b]);
      if constexpr (DET) atomicAdd(&GA[cell], fixed_from_product((double)(gwy * scale), (double)wx[b]));
      else atomicAdd(&GF[cell], gwy * wx[b]);
      sxv = fmaf(t, wx[b], sxv);
      sdx = fmaf(t, dwx[b], sdx);
    }
    gix = fmaf(wy[a], sdx, gix);
    giy = fmaf(dwy[a], sxv, giy);
    asm volatile("" ::: "memory");
  }
}

// exponent e of the fixed-point scale 2^(40 - e) for a bound with bit pattern `bits` on |cotangent|: |g| < 2^e;
// 255 = a non-finite cotangent (every sum becomes NaN), -128 = all zero so far
__device__ __forceinline__ int fixed_exponent(unsigned bits) {
  if (bits >= 0x7f800000u) return 255;
  if (bits == 0) return -128;
  int e = (int)(bits >> 23) - 126;                  // 2^(e-1) <= |g| < 2^e for normal numbers
  return e < -80 ? -80 : (e > 80 ? 80 : e);
}

template <int MODE, int R, bool DET>
__global__ void __launch_bounds__(STRIP_THREADS, STRIP_BWD_WAVES)
sl_advect_bwd_strip(const float* __restrict__ gout, const float* __restrict__ field, const float* __restrict__ u,
                    const float* __restrict__ v, float* __restrict__ gfield, float* __restrict__ gu,
                    float* __restrict__ gv, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                    const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans,
                    const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs, int64_t uv_bs,
                    int64_t gf_bs, int64_t guv_bs, int hx, int strips, unsigned long long* __restrict__ gacc,
                    const unsigned* __restrict__ pmax, unsigned* __restrict__ queue, unsigned* __restrict__ counts,
                    const unsigned* __restrict__ wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ unsigned qcount;
  // tall grids (round 5): the 32-row ring (63 KB: two workgroups per CU) for planes whose latitude displacements it
  // covers, the 64-row ring (126 KB: one) for the others - two launches back to back, a plane runs in the one its class
  // word names (adv_dy_class_kernel), the other returns at once; NULL = this launch takes every plane
  if (wide != nullptr && (wide[blockIdx.x / strips] != 0u) != (R == 64)) return;
  using Ring = StripRing<MODE, R>;
  constexpr int NT = Interp<MODE>::NT, RW = Ring::RW, RMASK = R - 1;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x / strips, strip = blockIdx.x - plane * strips;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;
  unsigned long long* GA = DET ? gacc + (int64_t)plane * P : nullptr;
  const int x0 = strip * STRIP_W, tw = min(STRIP_W, W - x0);
  const int WS = STRIP_W + 2 * hx + NT, wx0 = x0 + p - hx, nch = (WS + 63) >> 6;
  const int wn = R * WS;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);    // [R][WS] fixed-point sums
  float* ring = smem + 2 * wn;                                               // [R][WS] field window
  unsigned* stepmax = reinterpret_cast<unsigned*>(ring + wn);                // [3] max |cotangent| bits of a step
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const float gm0 = gmeans[2 * plane], gm1 = gmeans[2 * plane + 1];
  const StripLane tl(wave & 1, lane, x0, tw, W);
  const int rw = wave >> 1;
  const int nsteps = (H + STRIP_ROWS - 1) / STRIP_ROWS;
  const StripCols cs = strip_cols(wx0, WS, W, p, lane);
  unsigned* const q = queue + (size_t)blockIdx.x * 3 * (size_t)(H * STRIP_W);    // deferred points (see the forward)
  if (tid == 0) qcount = 0u;

  float qu[2], qv[2], qg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = min(rw + 4 * j, H - 1) * W + tl.cbase;
    qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx]; qg[j] = srow(GO + idx)[tl.lx];
  }
  const float lonc = lon_cells(srow(lon + tl.cbase)[tl.lx], g);

  for (int i = tid; i < wn; i += STRIP_THREADS) acc[i] = 0ull;
  if (tid < 3) stepmax[tid] = 0u;
  float pre[STRIP_CH], pre2[STRIP_CH];
  const int lo0 = Ring::lo(0, p);
  for (int r8 = 0; r8 < RW; r8 += 2 * STRIP_ROWS) {
    strip_load_row(F, lo0 + r8 + wave, H, W, p, cs, nch, pre);
    if (r8 + STRIP_ROWS < RW) strip_load_row(F, lo0 + r8 + STRIP_ROWS + wave, H, W, p, cs, nch, pre2);
    strip_store_row(ring, lo0 + r8 + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre);
    if (r8 + STRIP_ROWS < RW) strip_store_row(ring, lo0 + r8 + STRIP_ROWS + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre2);
  }
  strip_load_row(F, lo0 + RW + wave, H, W, p, cs, nch, pre);
  __syncthreads();
  // bound on |cotangent| of step 0 (the pole rows' means ride along: they are what pole rows scatter)
  {
    unsigned mb = max(abs_bits(gm0), abs_bits(gm1));
#pragma unroll
    for (int j = 0; j < 2; ++j) mb = max(mb, abs_bits(qg[j]));
    mb = wave_umax_lane63(mb);
    if (lane == 63) atomicMax(&stepmax[0], mb);
  }
  __syncthreads();

  int ecur = DET ? fixed_exponent(pmax[plane]) : -128;      // exponent of the ring's accumulators (wave-uniform)
  float scale = 0.f, inv_scale = 0.f;
  auto set_scale = [&](int e) {
    ecur = e;
    if (e == 255) { scale = 0.f; inv_scale = NAN; }
    else if (e == -128) { scale = 0.f; inv_scale = 0.f; }
    else { scale = ldexpf(1.0f, 40 - e); inv_scale = ldexpf(1.0f, e - 40); }
  };
  set_scale(ecur);

  // flush padded row pr of the ring (complete: no later arrival row reaches it) and clear its accumulators
  auto flush_row = [&](int pr) {
    const StripRow r = strip_row(pr, H, p);
    unsigned long long* arow = acc + (pr & RMASK) * WS;
    const double inv = (double)inv_scale;
#pragma unroll
    for (int i = 0; i < STRIP_CH; ++i)
      if (i < nch) {
        const int lc = (int)lane + 64 * i;
        if (lc < WS) {
          const long long sv = (long long)arow[lc];
          if (sv != 0) {
            arow[lc] = 0ull;
            if (r.valid) {
              const int cell = r.sr * W + (int)(r.mir ? (cs.pk[i] >> 16) : (cs.pk[i] & 0xffffu));
              if constexpr (DET) atomicAdd(&GA[cell], (unsigned long long)sv);
              else atomicAdd(&GF[cell], (float)((double)sv * inv));
            }
          }
        }
      }
  };

  const float Hpf = (float)Hp, Wpf = (float)Wp, Wf = (float)W, wx0f = (float)wx0, WSf = (float)WS;
  for (int s = 0; s < nsteps; ++s) {
    const int lo = Ring::lo(s, p);
    if constexpr (!DET) {
      // the scale follows the running maximum of |cotangent|: a step that raises it shifts every accumulator of the
      // ring down to the new exponent (rare: once per strip on smooth cotangents)
      const int e = fixed_exponent(stepmax[s % 3]);
      if (e > ecur) {
        if (ecur != -128 && e != 255) {
          const int sh = min(e - ecur, 63);
          for (int i = tid; i < wn; i += STRIP_THREADS) acc[i] = (unsigned long long)((long long)acc[i] >> sh);
        }
        set_scale(e);
        __syncthreads();
      }
      if (tid == 0) stepmax[(s + 2) % 3] = 0u;
    }
    // rows that left the window after step s - 1 go out; the rows entering at step s + 1 take their slots
    if (s > 0) flush_row(lo - STRIP_ROWS + wave);
    strip_store_row(ring, lo + RW + wave, RMASK, WS, H, p, m0, m1, lane, nch, pre);
    if (s + 1 < nsteps) strip_load_row(F, lo + R + wave, H, W, p, cs, nch, pre);
    const float rlof = (float)lo, rhif = (float)(lo + RW - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = STRIP_ROWS * s + rw + 4 * j;          // wave-uniform
      const float cu = qu[j], cv = qv[j], cgo = qg[j];
      {
        const int idx = min(y + STRIP_ROWS, H - 1) * W + tl.cbase;
        qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx]; qg[j] = srow(GO + idx)[tl.lx];
      }
      if (y < H) {
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty;
        DepState st;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, &st);
        int r0, c0;
        const bool fast = tap_block_ring<MODE>(ix, iy, Hpf, Wpf, Wf, wx0f, WSf, rlof, rhif, tx, ty, r0, c0);
        const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : cgo);
        float gix = 0.f, giy = 0.f;
        // (the sample coordinates ride along, as in the forward: the fix-up evaluates the gradient where the forward sampled)
        strip_defer(!fast && tl.active, (unsigned)(y * W + tl.cbase) + tl.lx, q, &qcount, lane, (size_t)H * STRIP_W, ix, iy);
        if (tl.active && fast) {
          {
            float wx[NT], wy[NT], dwx[NT], dwy[NT];
            Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
            Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
            const float gs_ = gval * scale;
            double wxd[NT];
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) wxd[bb] = (double)wx[bb];
#pragma unroll
            for (int a = 0; a < NT; ++a) {
              const int rb = ((r0 + a) & RMASK) * WS + c0;
              float sxv = 0.f, sdx = 0.f;
              const double gwy = (double)(gs_ * wy[a]);
#pragma unroll
              for (int bb = 0; bb < NT; ++bb) {
                const float val = ring[rb + bb];
                atomicAdd(&acc[rb + bb], fixed_from_product(gwy, wxd[bb]));
                sxv = fmaf(val, wx[bb], sxv);
                sdx = fmaf(val, dwx[bb], sdx);
              }
              gix = fmaf(wy[a], sdx, gix);
              giy = fmaf(dwy[a], sxv, giy);
            }
          }
          float guv, gvv;
          departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
          srow(GU + y * W + tl.cbase)[tl.lx] = guv;
          srow(GV + y * W + tl.cbase)[tl.lx] = gvv;
        }
      }
    }
    if constexpr (!DET) {
      // |cotangent| of the next step (prefetched above), published by the barrier
      unsigned mb = max(abs_bits(qg[0]), abs_bits(qg[1]));
      mb = wave_umax_lane63(mb);
      if (lane == 63 && s + 1 < nsteps) atomicMax(&stepmax[(s + 1) % 3], mb);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (raw: the stores and atomics stay in flight)
  }
  // what is still resident: the window of the last step
  const int lo_last = Ring::lo(nsteps - 1, p);
  for (int r8 = 0; r8 < RW; r8 += STRIP_ROWS) flush_row(lo_last + r8 + wave);
  if (tid == 0) counts[blockIdx.x] = qcount;
}

// Backward on grids of at most 256 longitudes (configs[3]): the ring spans the WHOLE latitude circle (window column lc
// holds image column lc mod W, the first NT columns repeated behind the last), one workgroup per plane.  No tap leaves
// the window in longitude - the polar rows, whose departure points lie anywhere on their circle, stay on the LDS path -
// and no other workgroup writes this plane: the rows leaving the ring are folded onto their source cells IN LDS (the
// wrap columns onto columns 0..NT-1, the rows beyond a pole onto their mirror rows, half a circle away) and go out as
// plain 256-byte stores.  Against the 128-column strips: no float atomics (the strips' flush ran at the chip's
// 1.3 TB/s atomic rate), no zero fill of the gradient, 1.02 instead of 1.19 window cells per point.
// 4 WPR waves: WPR 64-column segments per row, two rows per wave and step.
template <int MODE, int R, int WPR, bool FLDS>
__global__ void __launch_bounds__(256 * WPR)
sl_advect_bwd_circle(const float* __restrict__ gout, const float* __restrict__ field, const float* __restrict__ u,
                     const float* __restrict__ v, float* __restrict__ gfield, float* __restrict__ gu,
                     float* __restrict__ gv, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                     const float* __restrict__ lat_cells, const float* __restrict__ lon, const float* __restrict__ fmeans,
                     const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs, int64_t uv_bs,
                     int64_t gf_bs, int64_t guv_bs, int strips, unsigned* __restrict__ queue, unsigned* __restrict__ counts,
                     const unsigned* __restrict__ wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ unsigned qcount;
  using Ring = StripRing<MODE, R>;
  constexpr int NT = Interp<MODE>::NT, RW = Ring::RW, RMASK = R - 1, NTHREADS = 256 * WPR, SUBS = WPR / 2;
  const int H = g.H, W = g.W, p = g.p, P = H * W, Hp = H + 2 * p, Wp = W + 2 * p;
  const int tid = threadIdx.x;
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x;
  // two variants of this kernel are launched back to back; a plane runs in the one its displacement class names
  // (adv_dy_class_kernel: sampled max |v dt| in rows), the other one returns at once
  if ((wide[plane] != 0u) == FLDS) return;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;
  const int WS = W + NT, wx0 = p;
  const int wn = R * WS;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);    // [R][WS] fixed-point sums
  float* ring = smem + 2 * wn;                                               // [R][WS] field window (FLDS only)
  unsigned* stepmax = reinterpret_cast<unsigned*>(ring + (FLDS ? wn : 0));   // [3] max |cotangent| bits of a step
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const float gm0 = gmeans[2 * plane], gm1 = gmeans[2 * plane + 1];
  const StripLane tl(wave % WPR, lane, 0, W, W);
  const int rw = wave / WPR;                         // rows rw and rw + 4 of a step
  const int nsteps = (H + STRIP_ROWS - 1) / STRIP_ROWS;
  // staging / flushing: wave -> ring row (wave & 7), 64-column chunks [cfirst, cfirst + nch) of the window row
  const int srw = wave & 7, sub = wave >> 3;
  const int nch_all = (WS + 63) >> 6, per = (nch_all + SUBS - 1) / SUBS;
  const int cfirst = sub * per, nch = max(0, min(per, nch_all - cfirst));
  const StripCols cs = strip_cols(wx0, WS, W, p, lane, cfirst);
  unsigned* const q = queue + (size_t)plane * strips * 3 * (size_t)(H * STRIP_W);   // deferred points: |dy| beyond the ring
  if (tid == 0) qcount = 0u;

  float qu[2], qv[2], qg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = min(rw + 4 * j, H - 1) * W + tl.cbase;
    qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx]; qg[j] = srow(GO + idx)[tl.lx];
  }
  const float lonc = lon_cells(srow(lon + tl.cbase)[tl.lx], g);

  for (int i = tid; i < wn; i += NTHREADS) acc[i] = 0ull;
  if (tid < 3) stepmax[tid] = 0u;
  float pre[STRIP_CH];
  const int lo0 = Ring::lo(0, p);
  if constexpr (FLDS) {
    for (int r8 = 0; r8 < RW; r8 += STRIP_ROWS) {
      strip_load_row(F, lo0 + r8 + srw, H, W, p, cs, nch, pre);
      strip_store_row(ring, lo0 + r8 + srw, RMASK, WS, H, p, m0, m1, lane, nch, pre, cfirst);
    }
    strip_load_row(F, lo0 + RW + srw, H, W, p, cs, nch, pre);
  }
  __syncthreads();
  {
    unsigned mb = max(abs_bits(gm0), abs_bits(gm1));
#pragma unroll
    for (int j = 0; j < 2; ++j) mb = max(mb, abs_bits(qg[j]));
    mb = wave_umax_lane63(mb);
    if (lane == 63) atomicMax(&stepmax[0], mb);
  }
  __syncthreads();

  int ecur = -128;
  float scale = 0.f, inv_scale = 0.f;
  auto set_scale = [&](int e) {
    ecur = e;
    if (e == 255) { scale = 0.f; inv_scale = NAN; }
    else if (e == -128) { scale = 0.f; inv_scale = 0.f; }
    else { scale = ldexpf(1.0f, 40 - e); inv_scale = ldexpf(1.0f, e - 40); }
  };

  // a row beyond a pole: its sums belong to the mirror row, half a circle away, which is still in the ring
  auto fold_row = [&](int pr) {
    const StripRow r = strip_row(pr, H, p);
    if (!(r.valid && r.mir)) return;                       // wave-uniform
    unsigned long long* arow = acc + (pr & RMASK) * WS;
    unsigned long long* trow = acc + ((r.sr + p) & RMASK) * WS;
#pragma unroll
    for (int i = 0; i < STRIP_CH; ++i)
      if (i < nch) {
        const int lc = (int)lane + 64 * (cfirst + i);
        if (lc < WS) {
          const unsigned long long sv = arow[lc];
          if (sv != 0ull) {
            arow[lc] = 0ull;
            atomicAdd(&trow[cs.pk[i] >> 16], sv);
          }
        }
      }
  };
  // a source row leaves complete: wrap columns folded in, plain stores (every cell of the plane is written once)
  auto flush_row = [&](int pr) {
    const StripRow r = strip_row(pr, H, p);
    if (!r.valid || r.mir) return;                          // wave-uniform
    unsigned long long* arow = acc + (pr & RMASK) * WS;
    const double inv = (double)inv_scale;
    const global_ptr<float> dst = srow(GF + (int64_t)r.sr * W);
#pragma unroll
    for (int i = 0; i < STRIP_CH; ++i)
      if (i < nch) {
        const int j = (int)lane + 64 * (cfirst + i);
        if (j < W) {
          long long sv = (long long)arow[j];
          arow[j] = 0ull;
          if (j < NT) { sv += (long long)arow[j + W]; arow[j + W] = 0ull; }
          dst[j] = (float)((double)sv * inv);
        }
      }
  };

  const float Hpf = (float)Hp, Wpf = (float)Wp, Wf = (float)W, wx0f = (float)wx0, WSf = (float)WS;
  for (int s = 0; s < nsteps; ++s) {
    const int lo = Ring::lo(s, p);
    {
      const int e = fixed_exponent(stepmax[s % 3]);
      if (e > ecur) {
        if (ecur != -128 && e != 255) {
          const int sh = min(e - ecur, 63);
          for (int i = tid; i < wn; i += NTHREADS) acc[i] = (unsigned long long)((long long)acc[i] >> sh);
        }
        set_scale(e);
        __syncthreads();
      }
      if (tid == 0) stepmax[(s + 2) % 3] = 0u;
    }
    if (s > 0) {
      // (a row beyond the south pole and its mirror row never leave in the same group of eight: lo(s) = -5 mod 8)
      fold_row(lo - STRIP_ROWS + srw);
      flush_row(lo - STRIP_ROWS + srw);
    }
    if constexpr (FLDS) {
      strip_store_row(ring, lo + RW + srw, RMASK, WS, H, p, m0, m1, lane, nch, pre, cfirst);
      if (s + 1 < nsteps) strip_load_row(F, lo + R + srw, H, W, p, cs, nch, pre);
    }
    const float rlof = (float)lo, rhif = (float)(lo + RW - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = STRIP_ROWS * s + rw + 4 * j;          // wave-uniform
      const float cu = qu[j], cv = qv[j], cgo = qg[j];
      {
        const int idx = min(y + STRIP_ROWS, H - 1) * W + tl.cbase;
        qu[j] = srow(U + idx)[tl.lx]; qv[j] = srow(V + idx)[tl.lx]; qg[j] = srow(GO + idx)[tl.lx];
      }
      if (y < H) {
        const float sa = sin_lat[y * W], ca = cos_lat[y * W];
        float ix, iy, tx, ty;
        DepState st;
        departure_row(cu, cv, sa, ca, lonc, lat_cells[y * W], g, ix, iy, &st);
        int r0, c0;
        const bool fast = tap_block_ring<MODE>(ix, iy, Hpf, Wpf, Wf, wx0f, WSf, rlof, rhif, tx, ty, r0, c0);
        const float gval = (y == 0) ? gm0 : ((y == H - 1) ? gm1 : cgo);
        float gix = 0.f, giy = 0.f;
        // (the sample coordinates ride along, as in the forward: the fix-up evaluates the gradient where the forward sampled)
        strip_defer(!fast && tl.active, (unsigned)(y * W + tl.cbase) + tl.lx, q, &qcount, lane, (size_t)strips * H * STRIP_W, ix, iy);   // (one list per plane: three lists of strips x H x 128 entries)
        if (tl.active && fast) {
          float wx[NT], wy[NT], dwx[NT], dwy[NT];
          Interp<MODE>::weights(tx, wx); Interp<MODE>::weights(ty, wy);
          Interp<MODE>::dweights(tx, dwx); Interp<MODE>::dweights(ty, dwy);
          const float gs_ = gval * scale;
          double wxd[NT];
#pragma unroll
          for (int bb = 0; bb < NT; ++bb) wxd[bb] = (double)wx[bb];
          int fcol[NT], fcolm[NT];           // !FLDS: the field taps come from global memory (L1 / L2)
          if constexpr (!FLDS) {
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) {
              int j = c0 + bb;                // window column = image column (wx0 = p), wrap columns behind W
              j -= j >= W ? W : 0;
              int jm = j + (W >> 1);
              jm -= jm >= W ? W : 0;
              fcol[bb] = j; fcolm[bb] = jm;
            }
          }
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            const int rb = ((r0 + a) & RMASK) * WS + c0;
            float sxv = 0.f, sdx = 0.f;
            const double gwy = (double)(gs_ * wy[a]);
            float fval[NT];
            if constexpr (!FLDS) {
              bool mir;
              int pole;
              const int ro = tap_row(r0 + a, H, W, p, mir, pole);
#pragma unroll
              for (int bb = 0; bb < NT; ++bb) fval[bb] = F[ro + (mir ? fcolm[bb] : fcol[bb])];
#pragma unroll
              for (int bb = 0; bb < NT; ++bb) fval[bb] = pole == 1 ? m0 : (pole == 2 ? m1 : fval[bb]);
            }
#pragma unroll
            for (int bb = 0; bb < NT; ++bb) {
              const float val = FLDS ? ring[rb + bb] : fval[bb];
              atomicAdd(&acc[rb + bb], fixed_from_product(gwy, wxd[bb]));
              sxv = fmaf(val, wx[bb], sxv);
              sdx = fmaf(val, dwx[bb], sdx);
            }
            gix = fmaf(wy[a], sdx, gix);
            giy = fmaf(dwy[a], sxv, giy);
          }
          float guv, gvv;
          departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
          srow(GU + y * W + tl.cbase)[tl.lx] = guv;
          srow(GV + y * W + tl.cbase)[tl.lx] = gvv;
        }
      }
    }
    {
      unsigned mb = max(abs_bits(qg[0]), abs_bits(qg[1]));
      mb = wave_umax_lane63(mb);
      if (lane == 63 && s + 1 < nsteps) atomicMax(&stepmax[(s + 1) % 3], mb);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // what is still resident: first the rows beyond the poles onto their mirror rows, then the source rows
  const int lo_last = Ring::lo(nsteps - 1, p);
  for (int r8 = 0; r8 < RW; r8 += STRIP_ROWS) fold_row(lo_last + r8 + srw);
  __syncthreads();
  for (int r8 = 0; r8 < RW; r8 += STRIP_ROWS) flush_row(lo_last + r8 + srw);
  if (tid < strips) counts[(size_t)plane * strips + tid] = tid == 0 ? qcount : 0u;
}

// displacement class of a plane for the full-circle backward: 1 when more than 1/16 of the sampled points (every fourth
// row) move further in latitude than the ring of the field-in-LDS variant covers.  Such points are deferred to sixteen
// global float atomics each (~60 ps per point against ~14 in the ring): beyond ~9 % of them the 64-row variant, 35 %
// slower per point, wins.  A heuristic that only selects between two correct kernels.
__global__ void __launch_bounds__(256)
adv_dy_class_kernel(const float* __restrict__ v, unsigned* __restrict__ wide, int K, int H, int W, int64_t bs,
                    float thresh) {
  __shared__ unsigned red[4];
  const int plane = blockIdx.x, b = plane / K, k = plane - b * K;
  const float* V = v + (int64_t)b * bs + (int64_t)k * H * W;
  unsigned far = 0;
  for (int y = 0; y < H; y += 4)
    for (int x = threadIdx.x; x < W; x += 256) far += !(fabsf(V[y * W + x]) <= thresh) ? 1u : 0u;   // (a NaN counts as far)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) far += (unsigned)__shfl_xor((int)far, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = far;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned f = red[0] + red[1] + red[2] + red[3];
    const unsigned t = (unsigned)((H + 3) / 4) * (unsigned)W;      // points sampled
    wide[plane] = 16ull * f > t ? 1u : 0u;
  }
}

// the deferred points of the backward strips, one per lane: field gradient by global atomics, velocity gradients
template <int MODE, bool DET>
__global__ void __launch_bounds__(256)
sl_advect_bwd_strip_fixup(const float* __restrict__ gout, const float* __restrict__ field, const float* __restrict__ u,
                          const float* __restrict__ v, float* __restrict__ gfield, float* __restrict__ gu,
                          float* __restrict__ gv, const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                          const float* __restrict__ lon, const float* __restrict__ fmeans, const float* __restrict__ gmeans,
                          int K, AdvGeom g, int64_t go_bs, int64_t f_bs, int64_t uv_bs, int64_t gf_bs, int64_t guv_bs,
                          int strips, unsigned long long* __restrict__ gacc, const unsigned* __restrict__ pmax,
                          const unsigned* __restrict__ queue, const unsigned* __restrict__ counts, int lists_per_plane) {
  // lists_per_plane: `strips` (one list per strip, capacity H x 128) or 1 (the full-circle kernel: one list per plane
  // at the plane's first strip, capacity strips x H x 128)
  const int H = g.H, W = g.W, p = g.p, P = H * W;
  const unsigned n = counts[blockIdx.x];
  if (n == 0) return;
  const int plane = blockIdx.x / strips;
  const int b = plane / K, k = plane - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;
  unsigned long long* GA = DET ? gacc + (int64_t)plane * P : nullptr;
  const size_t qcap1 = (size_t)H * STRIP_W, qcap = lists_per_plane == 1 ? qcap1 * strips : qcap1;
  const unsigned* q = queue + (size_t)blockIdx.x * 3 * qcap1;
  const float* qx = reinterpret_cast<const float*>(q) + qcap;
  const float* qy = qx + qcap;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const float gm0 = gmeans[2 * plane], gm1 = gmeans[2 * plane + 1];
  float scale = 0.f;
  if constexpr (DET) {
    float inv;
    fixed_point_scale(__uint_as_float(pmax[plane]), scale, inv);
  }
  const unsigned last = (unsigned)(H - 1) * (unsigned)W;
  for (unsigned i = threadIdx.x; i < n; i += 256) {
    const unsigned idx = q[i];
    const float sa = sin_lat[idx], ca = cos_lat[idx];
    float ix_l, iy_l, gix, giy;
    DepState st;
    // the chain-rule state of the departure map, per lane (no wave-uniform branches: a list's order, and with it a
    // point's wave mates, varies from run to run).  The tap block and the weights are those of the coordinates the
    // strip kernel computed, decided on and stored - the ones the forward sampled at (departure_lane's own
    // coordinates differ from them in the last bits, which a polar point amplified to 2e-4 of its velocity gradient)
    departure_lane(U[idx], V[idx], sa, ca, lon_cells(lon[idx], g), g, ix_l, iy_l, &st);
    const float ix = qx[i], iy = qy[i];
    const float gval = idx < (unsigned)W ? gm0 : (idx >= last ? gm1 : GO[idx]);
    scatter_global<MODE, DET>(F, GF, GA, ix, iy, H, W, p, m0, m1, gval, scale, gix, giy);
    float guv, gvv;
    departure_backward(st, sa, ca, gix * gval, giy * gval, g, guv, gvv);
    GU[idx] = guv;
    GV[idx] = gvv;
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

// ---- deterministic tiled backward (PARADIS_DETERMINISTIC=1): per-plane max |cotangent| and the conversion of the
// integer accumulators; integer global atomics are associative, so the field gradient no longer depends on the order
// in which the tiles flush
__global__ void __launch_bounds__(256)
plane_absmax_kernel(const float* __restrict__ src, unsigned* __restrict__ pmax, int K, int P, int64_t bs) {
  __shared__ unsigned red[4];
  const int plane = blockIdx.x, b = plane / K, k = plane - b * K;
  const float* s = src + (int64_t)b * bs + (int64_t)k * P;
  unsigned m = 0;
  for (int i = threadIdx.x; i < P; i += 256) m = max(m, abs_bits(s[i]));
  m = wave_umax_lane63(m);
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) pmax[plane] = max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ void __launch_bounds__(256)
fixed_to_float_kernel(const unsigned long long* __restrict__ gacc, const unsigned* __restrict__ pmax,
                      float* __restrict__ dst, int P, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float scale, inv;
    fixed_point_scale(__uint_as_float(pmax[i / P]), scale, inv);
    dst[i] = (float)((double)(long long)gacc[i] * (double)inv);
  }
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
  PD_REQUIRE((int64_t)B * K < (1 << 30) && (int64_t)H * W < (1ll << 24), "%s: too large", name);
  return 0;
}

// constants of the coordinate map, evaluated in double from the reference's fp32 buffers
// (min_lat, min_lon, d_lat = max - min, d_lon: model/advection.py:67-72)
AdvGeom make_geom(int H, int W, int p, float dt, float min_lat, float min_lon, float d_lat, float d_lon) {
  AdvGeom g;
  g.H = H; g.W = W; g.p = p;
  g.ndt = -dt;
  const double cx = ((double)W - 1.0) / (double)d_lon, cy = ((double)H - 1.0) / (double)d_lat;
  const double per = 6.283185307179586476925286766559 * cx;
  g.cx = (float)cx; g.cy = (float)cy; g.cxd = cx;
  g.per = (float)per; g.inv_per = (float)(1.0 / per);
  g.c0xd = (double)p - (double)min_lon * cx;
  g.c0x = (float)g.c0xd;
  g.qoff = (float)(-g.c0xd / per);
  g.c0y = (float)((double)p - (double)min_lat * cy);
  return g;
}

constexpr size_t WHOLE_LDS_LIMIT = 64 * 1024;
// window halos (padded cells) of the tiled schedule.  Forward windows are cheap (4 B/cell); the
// backward holds 12 B/cell (64-bit accumulators + field), so its halo is what LDS allows at 2
// workgroups per CU.  Taps outside the window take the L2 / global-atomic path.
constexpr int HALO_FWD = 8;    // in-model optimum 6-12 at 128x256 and 721x1440; 24 pays only for ~45 px displacements
constexpr int HALO_BWD = ADV_HALO_BWD;   // two workgroups of 512 threads per CU
constexpr int MAX_HALO = 32;

template <typename K>
int reserve_lds(K kernel, const char* what, int bytes = 160 * 1024) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          bytes) != hipSuccess) {
    paradis_set_error(what);
    return 2;
  }
  return 0;
}

// `flags` of the C ABI (include/paradis_hip.h, PARADIS_ADVECT_*): schedule choice and window halo are
// per-call arguments, the library keeps no mutable state
size_t bwd_whole_lds(size_t cells) { return (3 * ((cells + 1) & ~(size_t)1) + 24) * sizeof(float); }   // 64-bit accumulators + field
bool use_tiled(size_t whole_bytes, int flags) {
  if (flags & PARADIS_ADVECT_TILED) return true;
  return whole_bytes > WHOLE_LDS_LIMIT;
}
int halo_of(int flags, int dflt, bool backward) {
  int h = (flags >> PARADIS_ADVECT_HALO_SHIFT) & 0xff;          // 0 = default, else halo + 1
  const int hb = (flags >> PARADIS_ADVECT_HALO_BWD_SHIFT) & 0xff;
  if (backward && hb) h = hb;
  return h == 0 ? dflt : std::min(h - 1, MAX_HALO);
}
// Does every tap block of a finite departure point lie inside a whole-plane window with `xr` extra
// columns?  ix = [0, period] + c0x, iy = [-pi/2, pi/2] cy + c0y, each with a margin for rounding; true for
// the global grids of the reference (period = W cells, latitudes from pole to pole).
bool taps_stay_inside(const AdvGeom& g, int NT, int xr) {
  const int off0 = NT == 4 ? -1 : 0;
  const double eps = 1e-2, hpi = 1.5707963267948966;
  const double x_lo = std::floor((double)g.c0x - eps) + off0, x_hi = std::floor((double)g.per + g.c0x + eps) + off0 + NT - 1;
  const double y_lo = std::floor(-hpi * g.cy + g.c0y - eps) + off0, y_hi = std::floor(hpi * g.cy + g.c0y + eps) + off0 + NT - 1;
  return x_lo >= 0 && x_hi <= g.W + 2 * g.p + xr - 1 && y_lo >= 0 && y_hi <= g.H + 2 * g.p - 1;
}
// one wave per latitude row with scalar table loads: W == 64 and a grid whose latitude depends on the
// row only and whose longitude depends on the column only (the caller vouches for it through `flags`)
bool separable(int flags, const float* lat_cells) {
  return (flags & PARADIS_ADVECT_SEPARABLE) && !(flags & PARADIS_ADVECT_GENERIC) && lat_cells != nullptr;
}
bool use_row64(int W, int flags, const float* lat_cells) { return W == 64 && separable(flags, lat_cells); }
// strip schedule: packed 16-bit source columns; ring rows by grid height (latitude halo 6 / 22 rows: displacements in
// the default model stay below 6 rows at 128 x 256 and below 22 rows for 99 % of the points at 721 x 1440,
// tools/adv_disp_stats.py); longitude halos: what 3 (forward) / 2 (backward) workgroups per CU leave room for
bool strip_ok(int W, int flags) { return W < 32768 && !(flags & PARADIS_ADVECT_TILES); }
int strip_ring_rows(int H) { return H <= 160 ? 32 : 64; }
int strip_halo_fwd(int W) { return W <= 512 ? 10 : 32; }
int strip_halo_bwd(int W) { return W <= 512 ? 10 : 16; }

}  // namespace

// pole-row means of field and cotangent (tiled schedule); in deterministic mode also the per-plane max |cotangent|
// and the 64-bit integer plane the tiles accumulate into
// layout: [pole means of field, cotangent: 4 floats per plane] [deterministic mode: per-plane max, integer plane]
// [strip schedule: counts and deferred-point lists, one H x 128 region per strip]
static size_t adv_ws_det_bytes(int B, int K, int H, int W) {
  return paradis_deterministic() ? (size_t)B * K * sizeof(unsigned) + 16 + (size_t)B * K * H * W * 8 + 16 : 0;
}
static size_t adv_ws_base_bytes(int B, int K) { return (((size_t)B * K * 4 * sizeof(float) + 255) & ~(size_t)255) + 256; }
static bool adv_uses_strips(int H, int W, int p, int flags) {
  // (the backward's whole-plane footprint is the larger one: sized for whichever direction tiles)
  return (flags & PARADIS_ADVECT_SEPARABLE) && !(flags & PARADIS_ADVECT_GENERIC) && strip_ok(W, flags) &&
         use_tiled(bwd_whole_lds((size_t)(H + 2 * p) * (W + 2 * p)), flags);
}
extern "C" size_t paradis_sl_advect_ws_bytes(int B, int K, int H, int W, int flags) {
  size_t n = adv_ws_base_bytes(B, K) + adv_ws_det_bytes(B, K, H, W) + 256;
  if (adv_uses_strips(H, W, 2, flags) || adv_uses_strips(H, W, 1, flags)) {
    const size_t strips = (size_t)B * K * ((W + STRIP_W - 1) / STRIP_W);
    n += 3 * strips * (size_t)H * STRIP_W * sizeof(unsigned) + (((strips + (size_t)B * K) * sizeof(unsigned) + 255) & ~(size_t)255) + 256;
  }
  return n;
}
// the deferred-point lists behind the other regions
static unsigned* adv_ws_queue(void* workspace, int B, int K, int H, int W, unsigned** counts) {
  char* base = (char*)workspace + adv_ws_base_bytes(B, K) + adv_ws_det_bytes(B, K, H, W);
  base = (char*)(((uintptr_t)base + 255) & ~(uintptr_t)255);
  const size_t strips = (size_t)B * K * ((W + STRIP_W - 1) / STRIP_W);
  *counts = (unsigned*)base;
  return (unsigned*)(base + (((strips + (size_t)B * K) * sizeof(unsigned) + 255) & ~(size_t)255));   // counts, class words, lists
}
inline int stream_blocks_adv(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 256 * 32); }

#define ADV_LAUNCH(KERNEL, WHOLE_, NTH_, grid, lds, ...)                                                  \
  do {                                                                                                    \
    if (mode == PARADIS_INTERP_BICUBIC)                                                                   \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BICUBIC, WHOLE_, NTH_>), dim3(grid), dim3(NTH_), lds, st,  \
                         __VA_ARGS__);                                                                    \
    else                                                                                                  \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BILINEAR, WHOLE_, NTH_>), dim3(grid), dim3(NTH_), lds, st, \
                         __VA_ARGS__);                                                                    \
  } while (0)

#define ADV_LAUNCH_ROW64(KERNEL, grid, lds, ...)                                                             \
  do {                                                                                                       \
    if (mode == PARADIS_INTERP_BICUBIC)                                                                      \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BICUBIC>), dim3(grid), dim3(256), lds, st, __VA_ARGS__);     \
    else                                                                                                     \
      hipLaunchKernelGGL((KERNEL<PARADIS_INTERP_BILINEAR>), dim3(grid), dim3(256), lds, st, __VA_ARGS__);    \
  } while (0)

extern "C" int paradis_sl_advect_fwd(const float* field, const float* u, const float* v, float* out,
                                     const float* sin_lat, const float* cos_lat, const float* lat_cells,
                                     const float* lon,
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
    if (use_row64(W, flags, lat_cells)) {
      const bool wide = taps_stay_inside(g, NT, ROW64_XR);
      const size_t lds = 2 * (size_t)(H + 2 * p) * (W + 2 * p + (wide ? ROW64_XR : 0)) * sizeof(float);
      static PerDeviceOnce once64;
      if (once64.first()) {
        if (reserve_lds(&sl_advect_fwd_row64<PARADIS_INTERP_BICUBIC, ROW64_XR>, "sl_advect_fwd: cannot reserve LDS") ||
            reserve_lds(&sl_advect_fwd_row64<PARADIS_INTERP_BICUBIC, 0>, "sl_advect_fwd: cannot reserve LDS") ||
            reserve_lds(&sl_advect_fwd_row64<PARADIS_INTERP_BILINEAR, ROW64_XR>, "sl_advect_fwd: cannot reserve LDS") ||
            reserve_lds(&sl_advect_fwd_row64<PARADIS_INTERP_BILINEAR, 0>, "sl_advect_fwd: cannot reserve LDS"))
          return 2;
      }
      const int chunk = ROW64_CHUNK, groups = (planes + chunk - 1) / chunk;
#define ROW64_FWD(MODE_, XR_)                                                                              \
      hipLaunchKernelGGL((sl_advect_fwd_row64<MODE_, XR_>), dim3(groups), dim3(256), lds, st, field, u, v, out, \
                         sin_lat, cos_lat, lat_cells, lon, K, g, f_bs, uv_bs, o_bs, planes, chunk)
      if (mode == PARADIS_INTERP_BICUBIC) { if (wide) ROW64_FWD(PARADIS_INTERP_BICUBIC, ROW64_XR); else ROW64_FWD(PARADIS_INTERP_BICUBIC, 0); }
      else { if (wide) ROW64_FWD(PARADIS_INTERP_BILINEAR, ROW64_XR); else ROW64_FWD(PARADIS_INTERP_BILINEAR, 0); }
#undef ROW64_FWD
    } else
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
#ifdef PARADIS_DEV_KNOBS
  static const int direct = [] { const char* e = getenv("PARADIS_ADVECT_DIRECT"); return e ? atoi(e) : 0; }();   // (A/B: 4 or 8 rows per workgroup)
  if (direct > 0 && separable(flags, lat_cells)) {
    const int drows = direct >= 8 ? 8 : 4, rgroups = (H + drows - 1) / drows, cblocks = (W + 63) / 64;
    const int64_t nwg = (int64_t)((planes + 7) / 8) * 8 * rgroups * cblocks;
    PD_REQUIRE(nwg < (1ll << 31), "sl_advect_fwd(direct): too many workgroups");
#define LAUNCH_DIRECT(M, D)                                                                                            \
    hipLaunchKernelGGL((sl_advect_fwd_direct<M, D>), dim3((unsigned)nwg), dim3(64 * D), 0, st, field, u, v, out, sin_lat,  \
                       cos_lat, lat_cells, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, planes, rgroups, cblocks)
    if (mode == PARADIS_INTERP_BICUBIC) { if (drows == 8) LAUNCH_DIRECT(PARADIS_INTERP_BICUBIC, 8); else LAUNCH_DIRECT(PARADIS_INTERP_BICUBIC, 4); }
    else { if (drows == 8) LAUNCH_DIRECT(PARADIS_INTERP_BILINEAR, 8); else LAUNCH_DIRECT(PARADIS_INTERP_BILINEAR, 4); }
#undef LAUNCH_DIRECT
  } else
#endif
  if (separable(flags, lat_cells) && strip_ok(W, flags)) {
    // strip schedule: ring of R padded rows, longitude halo hx (flags: PARADIS_ADVECT_HALO)
    const int ring = strip_ring_rows(H), hx = halo_of(flags, strip_halo_fwd(W), false);
    const int strips = (W + STRIP_W - 1) / STRIP_W;
    const size_t slds = (size_t)ring * (STRIP_W + 2 * hx + NT) * sizeof(float);
    PD_REQUIRE((int64_t)planes * strips < (1ll << 31), "sl_advect_fwd: too many strips");
    PD_REQUIRE(STRIP_W + 2 * hx + NT <= STRIP_MAX_WS, "sl_advect_fwd: halo too wide");
    static PerDeviceOnce once_row;
    if (once_row.first()) {
      if (reserve_lds(&sl_advect_fwd_strip<PARADIS_INTERP_BICUBIC, 32>, "sl_advect_fwd: cannot reserve LDS", STRIP_LDS_MAX) ||
          reserve_lds(&sl_advect_fwd_strip<PARADIS_INTERP_BILINEAR, 32>, "sl_advect_fwd: cannot reserve LDS", STRIP_LDS_MAX) ||
          reserve_lds(&sl_advect_fwd_strip<PARADIS_INTERP_BICUBIC, 64>, "sl_advect_fwd: cannot reserve LDS", STRIP_LDS_MAX) ||
          reserve_lds(&sl_advect_fwd_strip<PARADIS_INTERP_BILINEAR, 64>, "sl_advect_fwd: cannot reserve LDS", STRIP_LDS_MAX))
        return 2;
    }
#define LAUNCH_STRIP_FWD(M, R_)                                                                                   \
    hipLaunchKernelGGL((sl_advect_fwd_strip<M, R_>), dim3((unsigned)(planes * strips)), dim3(STRIP_THREADS), slds, st, \
                       field, u, v, out, sin_lat, cos_lat, lat_cells, lon, (const float*)fmeans, K, g, f_bs, uv_bs,   \
                       o_bs, hx, strips, queue, counts)
    unsigned* counts = nullptr;
    unsigned* queue = adv_ws_queue(workspace, B, K, H, W, &counts);
    if (mode == PARADIS_INTERP_BICUBIC) { if (ring == 32) LAUNCH_STRIP_FWD(PARADIS_INTERP_BICUBIC, 32); else LAUNCH_STRIP_FWD(PARADIS_INTERP_BICUBIC, 64); }
    else { if (ring == 32) LAUNCH_STRIP_FWD(PARADIS_INTERP_BILINEAR, 32); else LAUNCH_STRIP_FWD(PARADIS_INTERP_BILINEAR, 64); }
#undef LAUNCH_STRIP_FWD
    // the points whose tap block left the window, densely
    if (mode == PARADIS_INTERP_BICUBIC)
      hipLaunchKernelGGL((sl_advect_fwd_strip_fixup<PARADIS_INTERP_BICUBIC>), dim3((unsigned)(planes * strips)), dim3(256), 0, st,
                         field, u, v, out, sin_lat, cos_lat, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, strips,
                         (const unsigned*)queue, (const unsigned*)counts);
    else
      hipLaunchKernelGGL((sl_advect_fwd_strip_fixup<PARADIS_INTERP_BILINEAR>), dim3((unsigned)(planes * strips)), dim3(256), 0, st,
                         field, u, v, out, sin_lat, cos_lat, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, strips,
                         (const unsigned*)queue, (const unsigned*)counts);
  } else if (separable(flags, lat_cells)) {      // (diagnostic A/B: the tile schedule of rounds 2-3)
    static PerDeviceOnce once_row;
    if (once_row.first()) {
      if (reserve_lds(&sl_advect_fwd_tilerow<PARADIS_INTERP_BICUBIC>, "sl_advect_fwd: cannot reserve LDS") ||
          reserve_lds(&sl_advect_fwd_tilerow<PARADIS_INTERP_BILINEAR>, "sl_advect_fwd: cannot reserve LDS"))
        return 2;
    }
    if (mode == PARADIS_INTERP_BICUBIC)
      hipLaunchKernelGGL((sl_advect_fwd_tilerow<PARADIS_INTERP_BICUBIC>), dim3((unsigned)(planes * tiles)),
                         dim3(TILED_THREADS_FWD), lds, st, field, u, v, out, sin_lat, cos_lat, lat_cells,
                         lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, halo, tx, tiles);
    else
      hipLaunchKernelGGL((sl_advect_fwd_tilerow<PARADIS_INTERP_BILINEAR>), dim3((unsigned)(planes * tiles)),
                         dim3(TILED_THREADS_FWD), lds, st, field, u, v, out, sin_lat, cos_lat, lat_cells,
                         lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, halo, tx, tiles);
  } else {
    ADV_LAUNCH(sl_advect_fwd_kernel, false, TILED_THREADS_FWD, (unsigned)(planes * tiles), lds, field, u, v, out, sin_lat,
               cos_lat, lon, (const float*)fmeans, K, g, f_bs, uv_bs, o_bs, halo, tx, tiles, vec4);
  }
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, out, planes, K, H, W, o_bs);
  PD_CHECK_LAUNCH("sl_advect_fwd(tiled)");
  return 0;
}

extern "C" int paradis_sl_advect_bwd(const float* gout, const float* field, const float* u,
                                     const float* v, float* gfield, float* gu, float* gv,
                                     const float* sin_lat, const float* cos_lat, const float* lat_cells,
                                     const float* lon,
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
  auto lds_of = [](size_t cells) { return bwd_whole_lds(cells); };
  const int vec4 = (W % 4 == 0) && (f_bs % 4 == 0) && (((int64_t)H * W) % 4 == 0) && (p % 2 == 0) &&
                   (reinterpret_cast<uintptr_t>(field) & 15) == 0;
  const size_t whole = lds_of((size_t)(H + 2 * p) * (W + 2 * p));
  if (!use_tiled(whole, flags)) {
    if (use_row64(W, flags, lat_cells))
      ADV_LAUNCH_ROW64(sl_advect_bwd_row64, planes, whole, gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lat_cells, lon,
                       K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, vec4);
    else
      ADV_LAUNCH(sl_advect_bwd_kernel, true, 256, planes, whole, gout, field, u, v, gfield, gu, gv, sin_lat,
                 cos_lat, lon, (const float*)nullptr, (const float*)nullptr, K, g, go_bs, f_bs, uv_bs,
                 gf_bs, guv_bs, 0, 1, 1, vec4, (unsigned long long*)nullptr, (const unsigned*)nullptr);
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
  // PARADIS_DETERMINISTIC=1: the tiles add 64-bit fixed-point values (one power-of-two scale per plane) into an integer
  // plane in the workspace, converted once at the end - no float atomics on the field gradient
  unsigned* pmax = nullptr;
  unsigned long long* gacc = nullptr;
  if (paradis_deterministic()) {
    pmax = reinterpret_cast<unsigned*>(gmeans + (size_t)planes * 2);
    gacc = reinterpret_cast<unsigned long long*>(
        (reinterpret_cast<uintptr_t>(pmax + planes) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(plane_absmax_kernel, dim3(planes), dim3(256), 0, st, gout, pmax, K, P, go_bs);
  }
  // full-circle ring only where both of its variants fit LDS (H > 160 takes the 64-row ring: W beyond ~200 columns then
  // does not fit and the 128-column strips run instead, with their zero fill)
  const size_t circle_lds = ((size_t)strip_ring_rows(H) * (W + NT) * 3 + 8) * sizeof(float);
  const size_t circle_lds_wide = ((size_t)64 * (W + NT) * 2 + 8) * sizeof(float);
  const bool circle = separable(flags, lat_cells) && strip_ok(W, flags) && W <= 256 && !gacc && !(flags & PARADIS_ADVECT_STRIPS) &&
                      circle_lds <= (size_t)STRIP_LDS_MAX && circle_lds_wide <= (size_t)STRIP_LDS_MAX;
  if (!circle && pd_zero_async(gacc ? (void*)gacc : (void*)gfield, (size_t)planes * P * (gacc ? 8 : 4), st) != hipSuccess) {
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
    if (reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BICUBIC, false, TILED_THREADS_BWD, false>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BILINEAR, false, TILED_THREADS_BWD, false>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BICUBIC, false, TILED_THREADS_BWD, true>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_kernel<PARADIS_INTERP_BILINEAR, false, TILED_THREADS_BWD, true>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_tilerow<PARADIS_INTERP_BICUBIC, false>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_tilerow<PARADIS_INTERP_BILINEAR, false>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_tilerow<PARADIS_INTERP_BICUBIC, true>, "sl_advect_bwd: cannot reserve LDS") ||
        reserve_lds(&sl_advect_bwd_tilerow<PARADIS_INTERP_BILINEAR, true>, "sl_advect_bwd: cannot reserve LDS"))
      return 2;
  }
  const dim3 tgrid((unsigned)(planes * tiles)), tblock(TILED_THREADS_BWD);
#define TILED_BWD_ARGS(...) __VA_ARGS__, (const float*)fmeans, (const float*)gmeans, K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, halo, tx, tiles
#define LAUNCH_TILEROW(M, D)                                                                                     \
  hipLaunchKernelGGL((sl_advect_bwd_tilerow<M, D>), tgrid, tblock, lds, st,                                        \
                     TILED_BWD_ARGS(gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lat_cells, lon), gacc,    \
                     (const unsigned*)pmax)
#define LAUNCH_TILED(M, D)                                                                                       \
  hipLaunchKernelGGL((sl_advect_bwd_kernel<M, false, TILED_THREADS_BWD, D>), tgrid, tblock, lds, st,               \
                     TILED_BWD_ARGS(gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lon), vec4, gacc,         \
                     (const unsigned*)pmax)
  const bool cubic = mode == PARADIS_INTERP_BICUBIC;
  if (circle) {
    // full-circle ring, one workgroup per plane: plain stores (gfield needs no zero fill: the memset above is skipped).
    // Two variants, chosen per plane on the device (adv_dy_class_kernel): field + sums in a ring of strip_ring_rows(H)
    // rows, or - displacements of many rows - the sums alone in a ring of 64 rows with the field taps from L2.
    const int ring = strip_ring_rows(H), strips = (W + STRIP_W - 1) / STRIP_W;
    const size_t slds = circle_lds, slds_wide = circle_lds_wide;
    static PerDeviceOnce once_circle;
    if (once_circle.first()) {
#define RESERVE_CIRCLE(M, R_, WPR_, F_) reserve_lds(&sl_advect_bwd_circle<M, R_, WPR_, F_>, "sl_advect_bwd: cannot reserve LDS", STRIP_LDS_MAX)
#define RESERVE_CIRCLE_M(M)                                                                                         \
      (RESERVE_CIRCLE(M, 32, 2, true) || RESERVE_CIRCLE(M, 32, 4, true) || RESERVE_CIRCLE(M, 64, 2, true) ||        \
       RESERVE_CIRCLE(M, 64, 4, true) || RESERVE_CIRCLE(M, 64, 2, false) || RESERVE_CIRCLE(M, 64, 4, false))
      if (RESERVE_CIRCLE_M(PARADIS_INTERP_BICUBIC) || RESERVE_CIRCLE_M(PARADIS_INTERP_BILINEAR)) return 2;
#undef RESERVE_CIRCLE_M
#undef RESERVE_CIRCLE
    }
    unsigned* counts = nullptr;
    unsigned* queue = adv_ws_queue(workspace, B, K, H, W, &counts);
    unsigned* wide = counts + (size_t)planes * strips;      // class word per plane: behind the counts (adv_ws_queue)
    {
      // rows the field-in-LDS ring covers below an arrival row, in radians of v dt
      const int hyl = ring == 32 ? (mode == PARADIS_INTERP_BICUBIC ? 6 : 7) : (mode == PARADIS_INTERP_BICUBIC ? 22 : 23);
      const float thresh = ring == 64 ? INFINITY : (float)hyl / (g.cy * fabsf(dt));
      hipLaunchKernelGGL(adv_dy_class_kernel, dim3(planes), dim3(256), 0, st, v, wide, K, H, W, uv_bs, thresh);
    }
#define LAUNCH_CIRCLE(M, R_, WPR_, F_, LDS_)                                                                         \
    hipLaunchKernelGGL((sl_advect_bwd_circle<M, R_, WPR_, F_>), dim3((unsigned)planes), dim3(256 * WPR_), LDS_, st,    \
                       gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lat_cells, lon, (const float*)fmeans,      \
                       (const float*)gmeans, K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, strips, queue, counts,           \
                       (const unsigned*)wide)
#define LAUNCH_CIRCLE_M(M)                                                                                           \
    do {                                                                                                               \
      if (W <= 128) {                                                                                                  \
        if (ring == 32) LAUNCH_CIRCLE(M, 32, 2, true, slds); else LAUNCH_CIRCLE(M, 64, 2, true, slds);                 \
        if (ring == 32) LAUNCH_CIRCLE(M, 64, 2, false, slds_wide);                                                     \
      } else {                                                                                                         \
        if (ring == 32) LAUNCH_CIRCLE(M, 32, 4, true, slds); else LAUNCH_CIRCLE(M, 64, 4, true, slds);                 \
        if (ring == 32) LAUNCH_CIRCLE(M, 64, 4, false, slds_wide);                                                     \
      }                                                                                                                \
      hipLaunchKernelGGL((sl_advect_bwd_strip_fixup<M, false>), dim3((unsigned)(planes * strips)), dim3(256), 0, st,   \
                         gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lon, (const float*)fmeans,               \
                         (const float*)gmeans, K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, strips,                        \
                         (unsigned long long*)nullptr, (const unsigned*)nullptr, (const unsigned*)queue,               \
                         (const unsigned*)counts, 1);                                                                  \
    } while (0)
    if (cubic) LAUNCH_CIRCLE_M(PARADIS_INTERP_BICUBIC); else LAUNCH_CIRCLE_M(PARADIS_INTERP_BILINEAR);
#undef LAUNCH_CIRCLE_M
#undef LAUNCH_CIRCLE
  } else if (separable(flags, lat_cells) && strip_ok(W, flags)) {
    const int ring = strip_ring_rows(H), hx = halo_of(flags, strip_halo_bwd(W), true);
    const int strips = (W + STRIP_W - 1) / STRIP_W;
    auto slds_of = [&](int r) { return ((size_t)r * (STRIP_W + 2 * hx + NT) * 3 + 8) * sizeof(float); };
    const size_t slds = slds_of(ring);
    PD_REQUIRE(slds <= (size_t)STRIP_LDS_MAX, "sl_advect_bwd: window does not fit LDS");
    PD_REQUIRE((int64_t)planes * strips < (1ll << 31), "sl_advect_bwd: too many strips");
    PD_REQUIRE(STRIP_W + 2 * hx + NT <= STRIP_MAX_WS, "sl_advect_bwd: halo too wide");
    static PerDeviceOnce once_strip;
    if (once_strip.first()) {
#define RESERVE_STRIP(M, R_, D) reserve_lds(&sl_advect_bwd_strip<M, R_, D>, "sl_advect_bwd: cannot reserve LDS", STRIP_LDS_MAX)
      if (RESERVE_STRIP(PARADIS_INTERP_BICUBIC, 32, false) || RESERVE_STRIP(PARADIS_INTERP_BICUBIC, 32, true) ||
          RESERVE_STRIP(PARADIS_INTERP_BICUBIC, 64, false) || RESERVE_STRIP(PARADIS_INTERP_BICUBIC, 64, true) ||
          RESERVE_STRIP(PARADIS_INTERP_BILINEAR, 32, false) || RESERVE_STRIP(PARADIS_INTERP_BILINEAR, 32, true) ||
          RESERVE_STRIP(PARADIS_INTERP_BILINEAR, 64, false) || RESERVE_STRIP(PARADIS_INTERP_BILINEAR, 64, true))
        return 2;
#undef RESERVE_STRIP
    }
#define LAUNCH_STRIP_BWD(M, R_, D, WIDE_)                                                                           \
    hipLaunchKernelGGL((sl_advect_bwd_strip<M, R_, D>), dim3((unsigned)(planes * strips)), dim3(STRIP_THREADS),        \
                       slds_of(R_), st,                                                                                \
                       gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lat_cells, lon, (const float*)fmeans,      \
                       (const float*)gmeans, K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, hx, strips, gacc, (const unsigned*)pmax, \
                       queue, counts, (const unsigned*)(WIDE_))
    unsigned* wide = nullptr;       // class word per plane on tall grids (64-row ring), see sl_advect_bwd_strip
#define LAUNCH_STRIP_BWD_R(M, D)                                                                                     \
    do {                                                                                                               \
      if (ring == 32) LAUNCH_STRIP_BWD(M, 32, D, nullptr);                                                            \
      else { LAUNCH_STRIP_BWD(M, 32, D, wide); LAUNCH_STRIP_BWD(M, 64, D, wide); }                                    \
      hipLaunchKernelGGL((sl_advect_bwd_strip_fixup<M, D>), dim3((unsigned)(planes * strips)), dim3(256), 0, st, gout, \
                         field, u, v, gfield, gu, gv, sin_lat, cos_lat, lon, (const float*)fmeans, (const float*)gmeans, \
                         K, g, go_bs, f_bs, uv_bs, gf_bs, guv_bs, strips, gacc, (const unsigned*)pmax,                 \
                         (const unsigned*)queue, (const unsigned*)counts, strips);                                    \
    } while (0)
    unsigned* counts = nullptr;
    unsigned* queue = adv_ws_queue(workspace, B, K, H, W, &counts);
    if (ring == 64) {
      // planes whose sampled |v dt| stays inside the 32-row ring's latitude halo (more than 15/16 of the points) run in
      // it at two workgroups per CU: 721 x 1440 backward 20.7 -> 13.4 ms at small velocities (tools/advect_bench.py)
      wide = counts + (size_t)planes * strips;      // class word per plane: behind the counts (adv_ws_queue)
      const int hyl = mode == PARADIS_INTERP_BICUBIC ? 6 : 7;     // rows the 32-row ring covers below an arrival row
      hipLaunchKernelGGL(adv_dy_class_kernel, dim3(planes), dim3(256), 0, st, v, wide, K, H, W, uv_bs,
                         (float)hyl / (g.cy * fabsf(dt)));
    }
    if (gacc) { if (cubic) LAUNCH_STRIP_BWD_R(PARADIS_INTERP_BICUBIC, true); else LAUNCH_STRIP_BWD_R(PARADIS_INTERP_BILINEAR, true); }
    else { if (cubic) LAUNCH_STRIP_BWD_R(PARADIS_INTERP_BICUBIC, false); else LAUNCH_STRIP_BWD_R(PARADIS_INTERP_BILINEAR, false); }
#undef LAUNCH_STRIP_BWD_R
#undef LAUNCH_STRIP_BWD
  } else if (separable(flags, lat_cells)) {
    if (gacc) { if (cubic) LAUNCH_TILEROW(PARADIS_INTERP_BICUBIC, true); else LAUNCH_TILEROW(PARADIS_INTERP_BILINEAR, true); }
    else { if (cubic) LAUNCH_TILEROW(PARADIS_INTERP_BICUBIC, false); else LAUNCH_TILEROW(PARADIS_INTERP_BILINEAR, false); }
  } else {
    if (gacc) { if (cubic) LAUNCH_TILED(PARADIS_INTERP_BICUBIC, true); else LAUNCH_TILED(PARADIS_INTERP_BILINEAR, true); }
    else { if (cubic) LAUNCH_TILED(PARADIS_INTERP_BICUBIC, false); else LAUNCH_TILED(PARADIS_INTERP_BILINEAR, false); }
  }
#undef LAUNCH_TILED
#undef LAUNCH_TILEROW
#undef TILED_BWD_ARGS
  if (gacc)
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3(stream_blocks_adv((int64_t)planes * P)), dim3(256), 0, st,
                       (const unsigned long long*)gacc, (const unsigned*)pmax, gfield, P, (int64_t)planes * P);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, gfield, planes, K, H, W, gf_bs);
  PD_CHECK_LAUNCH("sl_advect_bwd(tiled)");
  return 0;
}
