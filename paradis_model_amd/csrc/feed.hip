// f4: the data feed on the device - temporal + top-of-atmosphere-radiation forcings
// (reference data/forcings/time_vars.py:6-40, data/forcings/toa_radiation.py:38-199, assembled as
// data/era5_dataset.py:587-621) and the feature normalisations (utils/normalization.py:6-80 as applied
// by data/era5_dataset.py:547-584).  The reference does this per sample with numpy in dataloader
// workers; here one launch per batch of timestamps.
//
// Arithmetic types follow the reference: solar geometry per instant in float64, the grid part in
// float32 with a float64 weight and float32 accumulation (numpy >= 2 promotion), compiled with
// -ffp-contract=off.  HBM traffic: 4 B per output value (write only); the kernel is bound by the
// 15 cosines per (time, cell).
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NQ = 15;
constexpr double JULIAN_REF_US = 946728000000000.0;  // 2000-01-01T12:00 in microseconds since 1970
// numpy.polynomial.legendre.leggauss(15)
__constant__ double QNODE[NQ] = {-0x1.f9da27c32e6d0p-1, -0x1.dfe24c4f8b448p-1, -0x1.b248221fffd63p-1, -0x1.72e6e181ab3c4p-1,
                                 -0x1.245676f08f3a4p-1, -0x1.939c69257d6b6p-2, -0x1.9c0ba62ef04b5p-3, 0x0.0p+0,
                                 0x1.9c0ba62ef04b5p-3,  0x1.939c69257d6b6p-2,  0x1.245676f08f3a4p-1,  0x1.72e6e181ab3c4p-1,
                                 0x1.b248221fffd63p-1,  0x1.dfe24c4f8b448p-1,  0x1.f9da27c32e6d0p-1};
__constant__ double QWEIGHT[NQ] = {0x1.f7dc7227a2aa8p-6, 0x1.2038260b5d022p-4, 0x1.b6ec9635f113ap-4, 0x1.1dd73b4963152p-3,
                                   0x1.5484f30a86eccp-3, 0x1.7d41fa76dc25bp-3, 0x1.96633f1fd02c2p-3, 0x1.9ee1575f9c972p-3,
                                   0x1.96633f1fd02c2p-3, 0x1.7d41fa76dc25bp-3, 0x1.5484f30a86eccp-3, 0x1.1dd73b4963152p-3,
                                   0x1.b6ec9635f113ap-4, 0x1.2038260b5d022p-4, 0x1.f7dc7227a2aa8p-6};

// numpy.mod for a positive divisor
__device__ __forceinline__ double pymod(double x, double d) {
  double r = fmod(x, d);
  if (r != 0.0 && r < 0.0) r += d;
  return r;
}

__device__ __forceinline__ int64_t floor_div(int64_t a, int64_t b) {
  int64_t q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

// days since 1970-01-01 of January 1st of the civil year containing day z (proleptic Gregorian)
__device__ int64_t year_start_days(int64_t z) {
  z += 719468;
  const int64_t era = floor_div(z, 146097);
  const int64_t doe = z - era * 146097;
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  int64_t y = yoe + era * 400;
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const int64_t mp = (5 * doy + 2) / 153;
  const int64_t m = mp < 10 ? mp + 3 : mp - 9;
  if (m <= 2) ++y;
  // days_from_civil(y, 1, 1)
  const int64_t yy = y - 1;   // January: year shifted by one in the March-based calendar
  const int64_t era2 = floor_div(yy, 400);
  const int64_t yoe2 = yy - era2 * 400;
  const int64_t doy2 = (153 * (1 + 9) + 2) / 5;   // Jan 1 -> month index 10 in the March-based year
  const int64_t doe2 = yoe2 * 365 + yoe2 / 4 - yoe2 / 100 + doy2;
  return era2 * 146097 + doe2 - 719468;
}

// per-instant scalars.  thread (t, q): q < 15 quadrature node of the hour ending at times[t];
// q == 15: the four temporal forcings of times[t].
//   sc_f[(t*15+q)*3 + {0,1,2}] = sin(decl), cos(decl), mod_day (float32);  sc_w[t*15+q] = weight (float64)
//   tf[t*4 + {0..3}] = sin/cos time of day, sin/cos year progress
__global__ void forcing_scalars_kernel(const int64_t* __restrict__ times_us, int T, float* __restrict__ tf,
                                       float* __restrict__ sc_f, double* __restrict__ sc_w) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = gid >> 4, q = gid & 15;
  if (t >= T) return;
  const int64_t us = times_us[t];
  if (q == NQ) {
    const int64_t hours = floor_div(us, 3600000000ll);          // astype("datetime64[h]")
    const int64_t days = floor_div(hours, 24);
    const double hour_of_day = (double)(hours - days * 24);
    const double tod = hour_of_day / 24.0;
    const double doy = (double)(hours - year_start_days(days) * 24) / 24.0;
    const double yp = doy / 365.25;
    const double two_pi = 2.0 * 3.141592653589793;
    tf[t * 4 + 0] = (float)sin(two_pi * tod);
    tf[t * 4 + 1] = (float)cos(two_pi * tod);
    tf[t * 4 + 2] = (float)sin(two_pi * yp);
    tf[t * 4 + 3] = (float)cos(two_pi * yp);
    return;
  }
  const double pi = 3.141592653589793;
  const double tq = (double)us - 3600e6 * (1.0 + QNODE[q]) / 2.0;
  const double mjd = (tq - JULIAN_REF_US) / 86400e6;
  const double anomaly = pymod(357.529 + 0.98560028 * mjd, 360.0) * pi / 180.0;
  const double mean_lon = pymod(280.459 + 0.98564736 * mjd, 360.0) * pi / 180.0;
  const double app_lon = mean_lon + (1.915 * sin(anomaly) + 0.020 * sin(2.0 * anomaly)) * pi / 180.0;
  const double dist = 1.00014 - 0.01671 * cos(anomaly) - 0.00014 * cos(2.0 * anomaly);
  const double obliq = (23.439 - 0.00000036 * mjd) * pi / 180.0;
  const double asc = atan2(cos(obliq) * sin(app_lon), cos(app_lon));
  const double decl = asin(sin(obliq) * sin(app_lon));
  const double eot = (pymod(mean_lon - asc + pi, 2.0 * pi) - pi) / (2.0 * pi);
  const float mod_day = (float)(pymod(mjd + eot, 1.0) * 2.0 * pi);
  const float decl32 = (float)decl;
  const int o = t * NQ + q;
  sc_f[o * 3 + 0] = sinf(decl32);
  sc_f[o * 3 + 1] = cosf(decl32);
  sc_f[o * 3 + 2] = mod_day;
  sc_w[o] = (1360.56 / (dist * dist)) * (3600.0 * QWEIGHT[q] / 2.0);
}

struct ForcingVars {
  int n;         // number of forcing variables
  int code[8];   // 0 = toa radiation, 1..4 = sin/cos time of day, sin/cos year progress
};

// thread (b, t, y, x): TOA radiation of one cell at one time of series b, scattered (with the temporal
// forcings of that time) to every window (s, k) with s + k = t:  out[b, s, y, x, v*n + k]
__global__ void __launch_bounds__(256)
forcings_kernel(const double* __restrict__ lat_deg, const double* __restrict__ lon_deg, int lat_f32, int B,
                int T, int H, int W, int n, ForcingVars fv, float toa_mean, float toa_std,
                const float* __restrict__ tf, const float* __restrict__ sc_f, const double* __restrict__ sc_w,
                float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t P = (int64_t)H * W;
  if (gid >= (int64_t)B * T * P) return;
  const int t = (int)(gid / P);            // flat (series, time) index
  const int64_t cell = gid - (int64_t)t * P;
  const int b = t / T, tl = t - b * T;     // windows never straddle two series
  const int y = (int)(cell / W), x = (int)(cell - (int64_t)y * W);
  bool need_toa = false;
  for (int v = 0; v < fv.n; ++v) need_toa |= fv.code[v] == 0;
  float toa = 0.f;
  if (need_toa) {
    // toa_radiation.py:135-136: lat*pi/180 in the latitude array's own dtype, then float32
    float lat_rad;
    if (lat_f32) lat_rad = ((float)lat_deg[y] * 3.14159274f) / 180.0f;
    else lat_rad = (float)(lat_deg[y] * 3.141592653589793 / 180.0);
    const float lon32 = (float)lon_deg[x];
    const float slat = sinf(lat_rad), clat = cosf(lat_rad);
    const float lon_rad = (lon32 * 3.14159274f) / 180.0f;
    float acc = 0.f;
    for (int q = 0; q < NQ; ++q) {
      const int o = t * NQ + q;
      const float sdec = sc_f[o * 3], cdec = sc_f[o * 3 + 1], mod_day = sc_f[o * 3 + 2];
      const float clst = cosf(lon_rad + mod_day);
      const float cz = fmaxf(0.f, slat * sdec + clat * cdec * clst);
      acc = (float)((double)acc + (double)cz * sc_w[o]);
    }
    toa = (acc - toa_mean) / toa_std;
  }
  const int steps = T - n + 1, C = fv.n * n;
  for (int k = 0; k < n; ++k) {
    const int s = tl - k;
    if (s < 0 || s >= steps) continue;
    float* o = out + (((int64_t)b * steps + s) * P + cell) * C;
    for (int v = 0; v < fv.n; ++v) o[v * n + k] = fv.code[v] == 0 ? toa : tf[t * 4 + fv.code[v] - 1];
  }
}

// channels-last feature (de)normalisation, in place: data[i, c], kind[c] in {0 none, 1 z-score (p0 mean,
// p1 std), 2 humidity (p0 q_min, p1 q_max), 3 precipitation}
__global__ void __launch_bounds__(256)
normalize_kernel(float* __restrict__ data, const int* __restrict__ kind, const float* __restrict__ p0,
                 const float* __restrict__ p1, int64_t n, int C, float eps_q, int inverse) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int k = kind[c];
  if (k == 0) return;
  const float x = data[i], a = p0[c], b = p1[c];
  float r = x;
  if (k == 1) {
    r = inverse ? x * b + a : (x - a) / b;
  } else if (k == 2) {
    const float lmin = logf(a), lmax = logf(b);
    if (!inverse) {
      r = (logf(fminf(fmaxf(x, 0.f), b) + eps_q) - lmin) / (lmax - lmin);
    } else {
      const float q = expf(x * (lmax - lmin) + lmin) - eps_q;
      r = fminf(fmaxf(q, 0.f), b);
    }
  } else if (k == 3) {
    r = inverse ? fmaxf(expf(x - 10.0f) - 1e-6f, 0.f) : logf(x + 1e-6f) + 10.0f;
  }
  data[i] = r;
}

}  // namespace

extern "C" size_t paradis_forcings_ws_bytes(int B, int T) {
  // per timestamp: [15] double weights, [15*3] float scalars, [4] float temporal forcings
  const size_t n = (size_t)(B < 0 ? 0 : B) * (size_t)(T < 0 ? 0 : T);
  return n * 15 * sizeof(double) + n * (15 * 3 + 4) * sizeof(float) + 256;
}

extern "C" int paradis_forcings(const int64_t* times_us, const double* lat_deg, const double* lon_deg,
                                int lat_is_f32, int B, int T, int H, int W, int n_time_inputs,
                                const int* var_codes, int n_vars, double toa_mean, double toa_std,
                                float* out, void* workspace, void* stream) {
  PD_REQUIRE(B >= 0 && T >= 1 && H >= 1 && W >= 1 && n_time_inputs >= 1 && n_time_inputs <= T,
             "forcings: bad shape B=%d T=%d H=%d W=%d n_time_inputs=%d", B, T, H, W, n_time_inputs);
  PD_REQUIRE((int64_t)B * T < (1 << 26), "forcings: too many timestamps");
  PD_REQUIRE(n_vars >= 1 && n_vars <= 8 && var_codes != nullptr, "forcings: 1..8 forcing variables");
  PD_REQUIRE(workspace != nullptr, "forcings: workspace required");
  ForcingVars fv{};
  fv.n = n_vars;
  for (int v = 0; v < n_vars; ++v) {
    PD_REQUIRE(var_codes[v] >= 0 && var_codes[v] <= 4, "forcings: unknown variable code %d", var_codes[v]);
    fv.code[v] = var_codes[v];
  }
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int TT = B * T;
  double* sc_w = (double*)workspace;
  float* sc_f = (float*)(sc_w + (size_t)TT * 15);
  float* tf = sc_f + (size_t)TT * 15 * 3;
  hipLaunchKernelGGL(forcing_scalars_kernel, dim3((TT * 16 + 63) / 64), dim3(64), 0, st, times_us, TT, tf, sc_f, sc_w);
  const int64_t total = (int64_t)TT * H * W;
  PD_REQUIRE((total + 255) / 256 < (1ll << 31), "forcings: too large");
  hipLaunchKernelGGL(forcings_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, lat_deg, lon_deg,
                     lat_is_f32, B, T, H, W, n_time_inputs, fv, (float)toa_mean, (float)toa_std, (const float*)tf,
                     (const float*)sc_f, (const double*)sc_w, out);
  PD_CHECK_LAUNCH("forcings");
  return 0;
}

extern "C" int paradis_normalize_features(float* data, const int* kind, const float* p0, const float* p1,
                                          int64_t rows, int C, float eps_q, int inverse, void* stream) {
  PD_REQUIRE(rows >= 0 && C >= 1, "normalize_features: bad shape");
  if (rows == 0) return 0;
  const int64_t n = rows * C;
  PD_REQUIRE((n + 255) / 256 < (1ll << 31), "normalize_features: too large");
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data,
                     kind, p0, p1, n, C, eps_q, inverse);
  PD_CHECK_LAUNCH("normalize_features");
  return 0;
}
