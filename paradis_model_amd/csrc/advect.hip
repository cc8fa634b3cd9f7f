// a3-a5: fused semi-Lagrangian advection core (reference model/advection.py:129-169).
//
//   F~  = F with rows 0, H-1 replaced by their longitudinal mean       (advection.py:100-114)
//   (phi, lam) departure point in the rotated frame                     (advection.py:74-98)
//   sample coordinates on the *virtual* geocyclic-padded plane          (advection.py:139-150 +
//       ATen grid_sampler unnormalise, align_corners=True)
//   bilinear / bicubic (Keys A=-0.75) gather through the a1 index map, zero outside the padded plane
//   rows 0, H-1 of the result replaced by their mean
//
// Saved-for-backward state is (F, u, v) only; everything else is recomputed.
// Algorithmic HBM traffic: 16 B/point forward, 28 B/point backward.
//
// Two schedules:
//   LDS  : one workgroup owns one (b,k) plane staged in LDS (planes <= 64 KiB forward,
//          2 planes <= 64 KiB backward, scatter-add through LDS float atomics);
//   GMEM : any plane size; taps are served by L2 / Infinity Cache, backward scatter uses
//          global float atomics; pole rows handled by small pre/post kernels.
//
// The coordinate chain keeps the reference's fp32 operation order; this file must be
// compiled with -ffp-contract=off (see Makefile) so that no extra FMAs are formed.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

#pragma clang fp contract(off)

// Diagnostic ablation switches (tools/advect_variants.py builds side libraries with them; the
// shipped library defines none):  ADV_NO_ATOMIC, ADV_NO_TRIG
namespace {

constexpr float TWO_PI_F = 6.283185307179586f;
constexpr float CLAMP_HI = 0.9999999f;  // float(1 - 1e-7), as torch.clamp converts its python bound
constexpr float KA = -0.75f;

struct AdvGeom {
  int H, W, p;
  float dt, min_lat, min_lon, d_lat, d_lon;
};

struct DepState {  // intermediates needed by the backward chain
  float sp, cp, sl, cl, s, n, d;
};

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
  sincosf(phi, &sp, &cp);
  sincosf(lam, &sl, &cl);
  const float cc = cp * cl;
  const float s = sp * ca + cc * sa;
  const float sc = fminf(fmaxf(s, -CLAMP_HI), CLAMP_HI);
  const float lat_d = asinf(sc);
  const float n = cp * sl;
  const float d = cc * ca - sp * sa;
  float lon_d = lon_a + atan2f(n, d);
  lon_d = lon_d + TWO_PI_F;
  float m = fmodf(lon_d, TWO_PI_F);
  if (m != 0.f && m < 0.f) m += TWO_PI_F;
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
struct TapSet {
  static constexpr int NT = (MODE == PARADIS_INTERP_BICUBIC) ? 4 : 2;
  static constexpr int OFF0 = (MODE == PARADIS_INTERP_BICUBIC) ? -1 : 0;
  int roff[NT];   // source row * W (0 when the tap row is outside the padded plane)
  int col[NT];    // wrapped source column
  int colm[NT];   // wrapped source column + W/2 (used by rows mirrored about a pole)
  bool rok[NT], cok[NT], mir[NT];
  float wx[NT], wy[NT];

  __device__ __forceinline__ void weights(float tx, float ty) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      wx[0] = cub2(tx + 1.f); wx[1] = cub1(tx); wx[2] = cub1(1.f - tx); wx[3] = cub2(2.f - tx);
      wy[0] = cub2(ty + 1.f); wy[1] = cub1(ty); wy[2] = cub1(1.f - ty); wy[3] = cub2(2.f - ty);
    } else {
      wx[0] = 1.f - tx; wx[1] = tx;
      wy[0] = 1.f - ty; wy[1] = ty;
    }
  }
  static __device__ __forceinline__ void dweights(float t, float* dw) {
    if (MODE == PARADIS_INTERP_BICUBIC) {
      dw[0] = dcub2(t + 1.f); dw[1] = dcub1(t); dw[2] = -dcub1(1.f - t); dw[3] = -dcub2(2.f - t);
    } else {
      dw[0] = -1.f; dw[1] = 1.f;
    }
  }

  __device__ __forceinline__ void setup(float ix, float iy, const AdvGeom& g, float& tx, float& ty) {
    const int H = g.H, W = g.W, p = g.p;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    float x0f = floorf(ix), y0f = floorf(iy);
    tx = ix - x0f;
    ty = iy - y0f;
    // non-finite or absurd coordinates: every tap is outside the padded plane (contributes 0)
    bool sane = (fabsf(ix) < 1e8f) && (fabsf(iy) < 1e8f);
    int x0 = sane ? (int)x0f : -1000000, y0 = sane ? (int)y0f : -1000000;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      int c = x0 + OFF0 + b;
      cok[b] = (c >= 0) && (c < Wp);
      int j = cok[b] ? geo_wrap_col(c - p, W) : 0;
      col[b] = j;
      int jm = j + (W >> 1);
      colm[b] = jm >= W ? jm - W : jm;
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      int r = y0 + OFF0 + a;
      rok[a] = (r >= 0) && (r < Hp);
      int ii = r - p, sr = 0;
      mir[a] = false;
      if (rok[a]) {
        if (ii < 0) { sr = -ii; mir[a] = true; }
        else if (ii >= H) { sr = 2 * (H - 1) - ii; mir[a] = true; }
        else sr = ii;
      }
      roff[a] = sr * W;
    }
    weights(tx, ty);
  }
  __device__ __forceinline__ int src(int a, int b) const { return roff[a] + (mir[a] ? colm[b] : col[b]); }
  __device__ __forceinline__ bool ok(int a, int b) const { return rok[a] && cok[b]; }
};

// ---- pole-row helpers ------------------------------------------------------------------
// mean over W of `row` (LDS or global), computed by one full wave
__device__ __forceinline__ float wave_row_mean(const float* row, int W) {
  float s = 0.f;
  for (int x = threadIdx.x & 63; x < W; x += 64) s += row[x];
  return wave_sum(s) / (float)W;
}

// ======================================================================================
// LDS schedule, forward
// ======================================================================================
template <int MODE>
__global__ void __launch_bounds__(256)
sl_advect_fwd_lds(const float* __restrict__ field, const float* __restrict__ u,
                  const float* __restrict__ v, float* __restrict__ out,
                  const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                  const float* __restrict__ lon, int K, AdvGeom g, int64_t f_bs, int64_t uv_bs,
                  int64_t o_bs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, W = g.W, P = H * W;
  float* Ft = smem;            // [P]
  float* pole_out = smem + P;  // [2*W]
  const int tid = threadIdx.x, wave = tid >> 6;
  const int b = blockIdx.x / K, k = blockIdx.x - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  float* O = out + (int64_t)b * o_bs + (int64_t)k * P;

  if ((P & 3) == 0 && ((reinterpret_cast<uintptr_t>(F) & 15) == 0)) {
    for (int i = tid * 4; i < P; i += 1024)
      *reinterpret_cast<float4*>(Ft + i) = *reinterpret_cast<const float4*>(F + i);
  } else {
    for (int i = tid; i < P; i += 256) Ft[i] = F[i];
  }
  __syncthreads();
  if (wave < 2) {
    float* row = Ft + (wave == 0 ? 0 : (H - 1) * W);
    float m = wave_row_mean(row, W);
    for (int x = tid & 63; x < W; x += 64) row[x] = m;
  }
  __syncthreads();

  for (int idx = tid; idx < P; idx += 256) {
    const int y = idx / W, x = idx - y * W;
    float ix, iy, tx, ty;
    departure(U[idx], V[idx], sin_lat[idx], cos_lat[idx], lon[idx], g, ix, iy, nullptr);
    TapSet<MODE> T;
    T.setup(ix, iy, g, tx, ty);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < TapSet<MODE>::NT; ++a) {
      float rowacc = 0.f;
#pragma unroll
      for (int bb = 0; bb < TapSet<MODE>::NT; ++bb) {
        float val = T.ok(a, bb) ? Ft[T.src(a, bb)] : 0.f;
        rowacc += val * T.wx[bb];
      }
      acc += rowacc * T.wy[a];
    }
    if (y == 0) pole_out[x] = acc;
    else if (y == H - 1) pole_out[W + x] = acc;
    else O[idx] = acc;
  }
  __syncthreads();
  if (wave < 2) {
    const float* row = pole_out + (wave == 0 ? 0 : W);
    float m = wave_row_mean(row, W);
    float* orow = O + (wave == 0 ? 0 : (int64_t)(H - 1) * W);
    for (int x = tid & 63; x < W; x += 64) orow[x] = m;
  }
}

// ======================================================================================
// LDS schedule, backward
//
// The field gradient is a scatter-add of 4x4 (2x2) weighted cotangents per point.  Measured on
// MI355X (tools/lds_atomic_bench.hip): ds_add_f32 costs ~195 cycles per wave-instruction (lanes are
// serialised) while ds_add_u64/u32 cost ~6-10.  The scatter therefore accumulates in 64-bit fixed
// point: every term is scaled by an exact power of two chosen from the plane's max |cotangent|
// (term < 2^41, resolution max|g| * 2^-41, far below fp32 epsilon), added with integer LDS atomics
// (associative => bitwise reproducible), and converted back once.
// ======================================================================================
template <int MODE>
__global__ void __launch_bounds__(256)
sl_advect_bwd_lds(const float* __restrict__ gout, const float* __restrict__ field,
                  const float* __restrict__ u, const float* __restrict__ v,
                  float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                  const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                  const float* __restrict__ lon, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                  int64_t uv_bs, int64_t gf_bs, int64_t guv_bs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = g.H, W = g.W, P = H * W;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem);  // [P] fixed-point sums
  float* Ft = smem + 2 * P;        // [P]
  float* misc = Ft + P;            // [0..1] pole means of gout, [2..5] per-wave max, [6] scale, [7] 1/scale
  const int tid = threadIdx.x, wave = tid >> 6;
  const int b = blockIdx.x / K, k = blockIdx.x - b * K;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float* U = u + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* V = v + (int64_t)b * uv_bs + (int64_t)k * P;
  const float* GO = gout + (int64_t)b * go_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  float* GU = gu + (int64_t)b * guv_bs + (int64_t)k * P;
  float* GV = gv + (int64_t)b * guv_bs + (int64_t)k * P;

  float gmax = 0.f;
  for (int i = tid; i < P; i += 256) {
    Ft[i] = F[i];
    acc[i] = 0ull;
    gmax = fmaxf(gmax, fabsf(GO[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
  if ((tid & 63) == 0) misc[2 + wave] = gmax;
  __syncthreads();
  if (wave < 2) {
    float* row = Ft + (wave == 0 ? 0 : (H - 1) * W);
    float m = wave_row_mean(row, W);
    for (int x = tid & 63; x < W; x += 64) row[x] = m;
  } else {
    // adjoint of the final pole mean: the cotangent of a pole row is its own row mean
    const float* row = GO + (wave == 2 ? 0 : (int64_t)(H - 1) * W);
    float m = wave_row_mean(row, W);
    if ((tid & 63) == 0) misc[wave - 2] = m;
  }
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(misc[2], misc[3]), fmaxf(misc[4], misc[5]));
    int e = 0;
    float scale = 0.f, inv = 0.f;
    if (mx > 0.f && mx < INFINITY) {
      frexpf(mx, &e);                       // mx < 2^e
      e = e < -80 ? -80 : (e > 80 ? 80 : e);
      scale = ldexpf(1.0f, 40 - e);
      inv = ldexpf(1.0f, e - 40);
    } else if (!(mx < INFINITY)) {
      inv = NAN;                            // non-finite cotangent: propagate NaN like float adds would
    }
    misc[6] = scale;
    misc[7] = inv;
  }
  __syncthreads();
  const float scale = misc[6];

  const float kx = ((float)W - 1.0f) / g.d_lon, ky = ((float)H - 1.0f) / g.d_lat;
  for (int idx = tid; idx < P; idx += 256) {
    const int y = idx / W;
    const float sa = sin_lat[idx], ca = cos_lat[idx];
    float ix, iy, tx, ty;
    DepState st;
    departure(U[idx], V[idx], sa, ca, lon[idx], g, ix, iy, &st);
    TapSet<MODE> T;
    T.setup(ix, iy, g, tx, ty);
    constexpr int NT = TapSet<MODE>::NT;
    float dwx[NT], dwy[NT];
    TapSet<MODE>::dweights(tx, dwx);
    TapSet<MODE>::dweights(ty, dwy);
    const float gval = (y == 0) ? misc[0] : ((y == H - 1) ? misc[1] : GO[idx]);
    const float gs_ = gval * scale;
    float gix = 0.f, giy = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      float sx = 0.f, sdx = 0.f;
      const float gwy = gs_ * T.wy[a];
#pragma unroll
      for (int bb = 0; bb < NT; ++bb) {
        if (T.ok(a, bb)) {
          const int s = T.src(a, bb);
          const float val = Ft[s];
#ifndef ADV_NO_ATOMIC
          const long long q = __float2ll_rn(gwy * T.wx[bb]);
          atomicAdd(&acc[s], (unsigned long long)q);
#else
          sx += gwy * 1e-30f;
#endif
          sx += val * T.wx[bb];
          sdx += val * dwx[bb];
        }
      }
      gix += T.wy[a] * sdx;
      giy += dwy[a] * sx;
    }
    gix *= gval;
    giy *= gval;
    // chain through pixel mapping, remainder (unit slope), atan2, asin(clamp)
    const float glam_c = gix * kx, gphi_c = giy * ky;
    const float sc = fminf(fmaxf(st.s, -CLAMP_HI), CLAMP_HI);
    const float gs = (st.s >= -CLAMP_HI && st.s <= CLAMP_HI) ? gphi_c / sqrtf(1.0f - sc * sc) : 0.f;
    const float den = st.n * st.n + st.d * st.d;
    const float gn = glam_c * st.d / den;
    const float gd = -glam_c * st.n / den;
    const float gphi = gs * (st.cp * ca - st.sp * st.cl * sa) + gn * (-st.sp * st.sl) +
                       gd * (-st.sp * st.cl * ca - st.cp * sa);
    const float glam = gs * (-st.cp * st.sl * sa) + gn * (st.cp * st.cl) +
                       gd * (-st.cp * st.sl * ca);
    GU[idx] = -g.dt * glam;
    GV[idx] = -g.dt * gphi;
  }
  __syncthreads();
  // fixed point -> float (reuse Ft as the gF~ plane), then the adjoint of the first pole mean
  const double inv = (double)misc[7];
  for (int i = tid; i < P; i += 256) Ft[i] = (float)((double)(long long)acc[i] * inv);
  __syncthreads();
  if (wave < 2) {
    float* row = Ft + (wave == 0 ? 0 : (H - 1) * W);
    float m = wave_row_mean(row, W);
    for (int x = tid & 63; x < W; x += 64) row[x] = m;
  }
  __syncthreads();
  for (int i = tid; i < P; i += 256) GF[i] = Ft[i];
}

// ======================================================================================
// GMEM schedule (any plane size)
// ======================================================================================
// rowmeans[plane][0/1] = mean of row 0 / H-1 of src plane
__global__ void __launch_bounds__(256)
pole_row_means(const float* __restrict__ src, float* __restrict__ means, int planes, int K, int H,
               int W, int64_t bs) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;  // one wave per (plane,row)
  if (w >= planes * 2) return;
  const int plane = w >> 1, which = w & 1;
  const int b = plane / K, k = plane - b * K;
  const float* row = src + (int64_t)b * bs + (int64_t)k * H * W + (which ? (int64_t)(H - 1) * W : 0);
  float m = wave_row_mean(row, W);
  if ((threadIdx.x & 63) == 0) means[w] = m;
}

// dst pole rows <- their own row mean (in place)
__global__ void __launch_bounds__(256)
pole_rows_to_mean(float* __restrict__ dst, int planes, int K, int H, int W, int64_t bs) {
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (w >= planes * 2) return;
  const int plane = w >> 1, which = w & 1;
  const int b = plane / K, k = plane - b * K;
  float* row = dst + (int64_t)b * bs + (int64_t)k * H * W + (which ? (int64_t)(H - 1) * W : 0);
  float m = wave_row_mean(row, W);
  for (int x = threadIdx.x & 63; x < W; x += 64) row[x] = m;
}

template <int MODE>
__global__ void __launch_bounds__(256)
sl_advect_fwd_gmem(const float* __restrict__ field, const float* __restrict__ u,
                   const float* __restrict__ v, float* __restrict__ out,
                   const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                   const float* __restrict__ lon, const float* __restrict__ fmeans, int K,
                   AdvGeom g, int64_t f_bs, int64_t uv_bs, int64_t o_bs) {
  const int H = g.H, W = g.W, P = H * W;
  const int plane = blockIdx.y;
  const int b = plane / K, k = plane - b * K;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const int lastrow = (H - 1) * W;
  float ix, iy, tx, ty;
  departure(u[(int64_t)b * uv_bs + (int64_t)k * P + idx], v[(int64_t)b * uv_bs + (int64_t)k * P + idx],
            sin_lat[idx], cos_lat[idx], lon[idx], g, ix, iy, nullptr);
  TapSet<MODE> T;
  T.setup(ix, iy, g, tx, ty);
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < TapSet<MODE>::NT; ++a) {
    float rowacc = 0.f;
#pragma unroll
    for (int bb = 0; bb < TapSet<MODE>::NT; ++bb) {
      float val = 0.f;
      if (T.ok(a, bb)) {
        val = F[T.src(a, bb)];
        if (T.roff[a] == 0) val = m0;
        else if (T.roff[a] == lastrow) val = m1;
      }
      rowacc += val * T.wx[bb];
    }
    acc += rowacc * T.wy[a];
  }
  out[(int64_t)b * o_bs + (int64_t)k * P + idx] = acc;
}

template <int MODE>
__global__ void __launch_bounds__(256)
sl_advect_bwd_gmem(const float* __restrict__ gout, const float* __restrict__ field,
                   const float* __restrict__ u, const float* __restrict__ v,
                   float* __restrict__ gfield, float* __restrict__ gu, float* __restrict__ gv,
                   const float* __restrict__ sin_lat, const float* __restrict__ cos_lat,
                   const float* __restrict__ lon, const float* __restrict__ fmeans,
                   const float* __restrict__ gmeans, int K, AdvGeom g, int64_t go_bs, int64_t f_bs,
                   int64_t uv_bs, int64_t gf_bs, int64_t guv_bs) {
  const int H = g.H, W = g.W, P = H * W;
  const int plane = blockIdx.y;
  const int b = plane / K, k = plane - b * K;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const int y = idx / W;
  const float* F = field + (int64_t)b * f_bs + (int64_t)k * P;
  float* GF = gfield + (int64_t)b * gf_bs + (int64_t)k * P;
  const float m0 = fmeans[2 * plane], m1 = fmeans[2 * plane + 1];
  const int lastrow = (H - 1) * W;
  const float sa = sin_lat[idx], ca = cos_lat[idx];
  float ix, iy, tx, ty;
  DepState st;
  departure(u[(int64_t)b * uv_bs + (int64_t)k * P + idx], v[(int64_t)b * uv_bs + (int64_t)k * P + idx],
            sa, ca, lon[idx], g, ix, iy, &st);
  TapSet<MODE> T;
  T.setup(ix, iy, g, tx, ty);
  constexpr int NT = TapSet<MODE>::NT;
  float dwx[NT], dwy[NT];
  TapSet<MODE>::dweights(tx, dwx);
  TapSet<MODE>::dweights(ty, dwy);
  const float gval = (y == 0) ? gmeans[2 * plane]
                              : ((y == H - 1) ? gmeans[2 * plane + 1]
                                              : gout[(int64_t)b * go_bs + (int64_t)k * P + idx]);
  float gix = 0.f, giy = 0.f;
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float sx = 0.f, sdx = 0.f;
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) {
      if (T.ok(a, bb)) {
        const int s = T.src(a, bb);
        float val = F[s];
        if (T.roff[a] == 0) val = m0;
        else if (T.roff[a] == lastrow) val = m1;
        atomicAdd(&GF[s], gval * T.wy[a] * T.wx[bb]);
        sx += val * T.wx[bb];
        sdx += val * dwx[bb];
      }
    }
    gix += T.wy[a] * sdx;
    giy += dwy[a] * sx;
  }
  gix *= gval;
  giy *= gval;
  const float kx = ((float)W - 1.0f) / g.d_lon, ky = ((float)H - 1.0f) / g.d_lat;
  const float glam_c = gix * kx, gphi_c = giy * ky;
  const float sc = fminf(fmaxf(st.s, -CLAMP_HI), CLAMP_HI);
  const float gs = (st.s >= -CLAMP_HI && st.s <= CLAMP_HI) ? gphi_c / sqrtf(1.0f - sc * sc) : 0.f;
  const float den = st.n * st.n + st.d * st.d;
  const float gn = glam_c * st.d / den;
  const float gd = -glam_c * st.n / den;
  const float gphi = gs * (st.cp * ca - st.sp * st.cl * sa) + gn * (-st.sp * st.sl) +
                     gd * (-st.sp * st.cl * ca - st.cp * sa);
  const float glam = gs * (-st.cp * st.sl * sa) + gn * (st.cp * st.cl) + gd * (-st.cp * st.sl * ca);
  gu[(int64_t)b * guv_bs + (int64_t)k * P + idx] = -g.dt * glam;
  gv[(int64_t)b * guv_bs + (int64_t)k * P + idx] = -g.dt * gphi;
}

int check_adv(const char* name, int B, int K, int H, int W, int mode) {
  PD_REQUIRE(B >= 0 && K >= 1 && H >= 4 && W >= 4, "%s: bad shape B=%d K=%d H=%d W=%d", name, B, K, H, W);
  PD_REQUIRE(W % 2 == 0, "%s: Number of longitude points must be even", name);
  PD_REQUIRE(mode == PARADIS_INTERP_BILINEAR || mode == PARADIS_INTERP_BICUBIC,
             "%s: interpolation mode must be 1 (bilinear) or 2 (bicubic)", name);
  PD_REQUIRE((int64_t)B * K < (1 << 30) && (int64_t)H * W < (1ll << 30), "%s: too large", name);
  return 0;
}

constexpr size_t LDS_PLANE_LIMIT = 64 * 1024;

}  // namespace

// test hook: force the GMEM schedule regardless of plane size (set via env PARADIS_ADVECT_FORCE_GMEM)
static bool force_gmem() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PARADIS_ADVECT_FORCE_GMEM");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
extern "C" void paradis_debug_set_advect_gmem(int on);
static int g_force_gmem_override = -1;
extern "C" void paradis_debug_set_advect_gmem(int on) { g_force_gmem_override = on; }
static bool use_gmem(size_t lds_bytes) {
  if (g_force_gmem_override >= 0) return g_force_gmem_override == 1 || lds_bytes > LDS_PLANE_LIMIT;
  return force_gmem() || lds_bytes > LDS_PLANE_LIMIT;
}

extern "C" size_t paradis_sl_advect_ws_bytes(int B, int K, int H, int W) {
  (void)H; (void)W;
  return (size_t)B * K * 4 * sizeof(float) + 256;
}

extern "C" int paradis_sl_advect_fwd(const float* field, const float* u, const float* v, float* out,
                                     const float* sin_lat, const float* cos_lat, const float* lon,
                                     int B, int K, int H, int W, int64_t f_bs, int64_t uv_bs,
                                     int64_t o_bs, float dt, float min_lat, float min_lon,
                                     float d_lat, float d_lon, int mode, void* workspace,
                                     void* stream) {
  if (int e = check_adv("sl_advect_fwd", B, K, H, W, mode)) return e;
  if (B == 0) return 0;
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1;
  AdvGeom g{H, W, p, dt, min_lat, min_lon, d_lat, d_lon};
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K, P = H * W;
  const size_t lds = ((size_t)P + 2 * W) * sizeof(float);
  if (!use_gmem(lds)) {
    if (mode == PARADIS_INTERP_BICUBIC)
      hipLaunchKernelGGL(sl_advect_fwd_lds<PARADIS_INTERP_BICUBIC>, dim3(planes), dim3(256), lds, st,
                         field, u, v, out, sin_lat, cos_lat, lon, K, g, f_bs, uv_bs, o_bs);
    else
      hipLaunchKernelGGL(sl_advect_fwd_lds<PARADIS_INTERP_BILINEAR>, dim3(planes), dim3(256), lds, st,
                         field, u, v, out, sin_lat, cos_lat, lon, K, g, f_bs, uv_bs, o_bs);
    PD_CHECK_LAUNCH("sl_advect_fwd_lds");
    return 0;
  }
  PD_REQUIRE(workspace != nullptr, "sl_advect_fwd: workspace required for the GMEM schedule");
  PD_REQUIRE(planes <= 65535 * 1, "sl_advect_fwd: GMEM schedule supports at most 65535 planes per call");
  float* fmeans = (float*)workspace;
  const int mean_blocks = (planes * 2 * 64 + 255) / 256;
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, field, fmeans, planes, K, H, W, f_bs);
  dim3 grid((P + 255) / 256, planes);
  if (mode == PARADIS_INTERP_BICUBIC)
    hipLaunchKernelGGL(sl_advect_fwd_gmem<PARADIS_INTERP_BICUBIC>, grid, dim3(256), 0, st, field, u, v,
                       out, sin_lat, cos_lat, lon, fmeans, K, g, f_bs, uv_bs, o_bs);
  else
    hipLaunchKernelGGL(sl_advect_fwd_gmem<PARADIS_INTERP_BILINEAR>, grid, dim3(256), 0, st, field, u, v,
                       out, sin_lat, cos_lat, lon, fmeans, K, g, f_bs, uv_bs, o_bs);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, out, planes, K, H, W, o_bs);
  PD_CHECK_LAUNCH("sl_advect_fwd_gmem");
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
  const int p = mode == PARADIS_INTERP_BICUBIC ? 2 : 1;
  AdvGeom g{H, W, p, dt, min_lat, min_lon, d_lat, d_lon};
  hipStream_t st = (hipStream_t)stream;
  const int planes = B * K, P = H * W;
  const size_t lds = ((size_t)3 * P + 8) * sizeof(float);
  if (!use_gmem(lds)) {
    if (mode == PARADIS_INTERP_BICUBIC)
      hipLaunchKernelGGL(sl_advect_bwd_lds<PARADIS_INTERP_BICUBIC>, dim3(planes), dim3(256), lds, st,
                         gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lon, K, g, go_bs, f_bs,
                         uv_bs, gf_bs, guv_bs);
    else
      hipLaunchKernelGGL(sl_advect_bwd_lds<PARADIS_INTERP_BILINEAR>, dim3(planes), dim3(256), lds, st,
                         gout, field, u, v, gfield, gu, gv, sin_lat, cos_lat, lon, K, g, go_bs, f_bs,
                         uv_bs, gf_bs, guv_bs);
    PD_CHECK_LAUNCH("sl_advect_bwd_lds");
    return 0;
  }
  PD_REQUIRE(workspace != nullptr, "sl_advect_bwd: workspace required for the GMEM schedule");
  PD_REQUIRE(planes <= 65535, "sl_advect_bwd: GMEM schedule supports at most 65535 planes per call");
  PD_REQUIRE(gf_bs == (int64_t)K * P, "sl_advect_bwd: GMEM schedule needs a contiguous gfield");
  float* fmeans = (float*)workspace;
  float* gmeans = fmeans + (size_t)planes * 2;
  const int mean_blocks = (planes * 2 * 64 + 255) / 256;
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, field, fmeans, planes, K, H, W, f_bs);
  hipLaunchKernelGGL(pole_row_means, dim3(mean_blocks), dim3(256), 0, st, gout, gmeans, planes, K, H, W, go_bs);
  if (hipMemsetAsync(gfield, 0, (size_t)planes * P * sizeof(float), st) != hipSuccess) {
    paradis_set_error("sl_advect_bwd: memset failed");
    return 2;
  }
  dim3 grid((P + 255) / 256, planes);
  if (mode == PARADIS_INTERP_BICUBIC)
    hipLaunchKernelGGL(sl_advect_bwd_gmem<PARADIS_INTERP_BICUBIC>, grid, dim3(256), 0, st, gout, field,
                       u, v, gfield, gu, gv, sin_lat, cos_lat, lon, fmeans, gmeans, K, g, go_bs, f_bs,
                       uv_bs, gf_bs, guv_bs);
  else
    hipLaunchKernelGGL(sl_advect_bwd_gmem<PARADIS_INTERP_BILINEAR>, grid, dim3(256), 0, st, gout, field,
                       u, v, gfield, gu, gv, sin_lat, cos_lat, lon, fmeans, gmeans, K, g, go_bs, f_bs,
                       uv_bs, gf_bs, guv_bs);
  hipLaunchKernelGGL(pole_rows_to_mean, dim3(mean_blocks), dim3(256), 0, st, gfield, planes, K, H, W, gf_bs);
  PD_CHECK_LAUNCH("sl_advect_bwd_gmem");
  return 0;
}
