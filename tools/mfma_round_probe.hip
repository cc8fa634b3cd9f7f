// Diagnostic: how does v_mfma_f32_32x32x16_bf16 (and the f32 MFMA the "exact" GEMMs use) round?
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_round_probe.hip -o build/tools/mfma_round_probe && build/tools/mfma_round_probe
// Every output element of one MFMA is D = C + sum_k a_k b_k with the same (a_k, b_k, C) in every row / column, so no
// knowledge of the fragment layout is needed.  Cases:
//   1. C = 1, one product 1.5 * 2^-24 (1.5 half-ulps of C): round-to-nearest gives 1 + 2^-23, truncation 1.
//   2. C = 1, one product -2^-26: round-to-nearest gives 1, truncation towards zero 1 - 2^-24.
//   3. C = 1, sixteen products 2^-25: an exact 16-term sum gives 1 + 2^-21; sixteen individually rounded adds give 1.
//   4. C = 1, products 2^-24 and 2^-40: the tie is broken upwards only if the small product survives the alignment.
//   5. statistics: random bf16 operands, random C; signed error of D against the exact value in units of ulp(D),
//      mean and rms, for 1 MFMA and for a chain of 64 MFMAs (K = 1024) - a truncating accumulate shows as a mean of
//      about -0.5 ulp per accumulate in the direction of zero, round-to-nearest as mean 0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

static unsigned short bf16_bits(float x) {   // x must be representable
  uint32_t u;
  __builtin_memcpy(&u, &x, 4);
  return (unsigned short)(u >> 16);
}
static float bf16_val(unsigned short b) {
  uint32_t u = (uint32_t)b << 16;
  float x;
  __builtin_memcpy(&x, &u, 4);
  return x;
}

// one wave: nsteps chained MFMAs; step t uses a[t*16 + k], b[t*16 + k] (k = 0..15) in every row / column
__global__ void chain_bf16(const unsigned short* a, const unsigned short* b, float c0, int nsteps, float* out) {
  const int lane = threadIdx.x & 63, kh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  for (int t = 0; t < nsteps; ++t) {
    u16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
      av[j] = a[t * 16 + kh * 8 + j];
      bv[j] = b[t * 16 + kh * 8 + j];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
  }
  if (lane == 0) out[0] = acc[0];
  if (lane == 37) out[1] = acc[5];
}

// the f32 MFMA: k = 2 per instruction (lane>>5 = k), same-everywhere operands
__global__ void chain_f32(const float* a, const float* b, float c0, int nsteps, float* out) {
  const int lane = threadIdx.x & 63, kh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0;
  for (int t = 0; t < nsteps; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t * 2 + kh], b[t * 2 + kh], acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
  if (lane == 37) out[1] = acc[5];
}

// many independent problems at once for the statistics: problem = blockIdx.x * 4 + wave
__global__ void chain_bf16_many(const unsigned short* a, const unsigned short* b, const float* c0, int nsteps, float* out) {
  const int lane = threadIdx.x & 63, kh = lane >> 5, prob = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned short* pa = a + (size_t)prob * nsteps * 16;
  const unsigned short* pb = b + (size_t)prob * nsteps * 16;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0[prob];
  for (int t = 0; t < nsteps; ++t) {
    u16x8 av, bv;
    for (int j = 0; j < 8; ++j) {
      av[j] = pa[t * 16 + kh * 8 + j];
      bv[j] = pb[t * 16 + kh * 8 + j];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
  }
  if (lane == 0) out[prob] = acc[0];
}

__global__ void chain_f32_many(const float* a, const float* b, const float* c0, int nsteps, float* out) {
  const int lane = threadIdx.x & 63, kh = lane >> 5, prob = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const float* pa = a + (size_t)prob * nsteps * 2;
  const float* pb = b + (size_t)prob * nsteps * 2;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c0[prob];
  for (int t = 0; t < nsteps; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[t * 2 + kh], pb[t * 2 + kh], acc, 0, 0, 0);
  if (lane == 0) out[prob] = acc[0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float run_case(const std::vector<float>& a, const std::vector<float>& b, float c0, const char* name, double expect_rn,
                      double expect_rz) {
  const int n = (int)a.size(), steps = n / 16;
  std::vector<unsigned short> ha(n), hb(n);
  for (int i = 0; i < n; ++i) { ha[i] = bf16_bits(a[i]); hb[i] = bf16_bits(b[i]); }
  unsigned short *da, *db;
  float* dout;
  CK(hipMalloc(&da, n * 2)); CK(hipMalloc(&db, n * 2)); CK(hipMalloc(&dout, 8));
  CK(hipMemcpy(da, ha.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice));
  chain_bf16<<<1, 64>>>(da, db, c0, steps, dout);
  float out[2];
  CK(hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost));
  printf("%-58s got %.10e (%a)  round-to-nearest %.10e  truncation %.10e  [other lane %a]\n", name, out[0], out[0], expect_rn,
         expect_rz, out[1]);
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
  return out[0];
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
  const float u = ldexpf(1.f, -24);     // half an ulp of 1.0
  {
    std::vector<float> a(16, 0.f), b(16, 0.f);
    a[0] = 1.f; b[0] = 1.5f * u;
    run_case(a, b, 1.f, "1. C=1 + 1.5*2^-24", 1.0 + 2.0 * u, 1.0);
    b[0] = -0.25f * u;
    run_case(a, b, 1.f, "2. C=1 - 2^-26", 1.0, 1.0 - u);
    b[0] = 0.5f * u; a[0] = 1.f;
    for (int k = 0; k < 16; ++k) { a[k] = 1.f; b[k] = 0.5f * u; }
    run_case(a, b, 1.f, "3. C=1 + 16 x 2^-25 (exact sum 1 + 2^-21)", 1.0 + 16 * 0.5 * u, 1.0);
    for (int k = 0; k < 16; ++k) { a[k] = 0.f; b[k] = 0.f; }
    a[0] = 1.f; b[0] = u; a[9] = 1.f; b[9] = ldexpf(1.f, -40);
    run_case(a, b, 1.f, "4. C=1 + 2^-24 + 2^-40 (tie broken by a tiny product)", 1.0 + 2 * u, 1.0);
    a[9] = 0.f;
    run_case(a, b, 1.f, "4b. C=1 + 2^-24 (exact tie; nearest-even gives 1)", 1.0, 1.0);
    b[0] = 3.f * u;
    run_case(a, b, 1.f, "4c. C=1 + 3*2^-24 (exact tie; nearest-even gives 1+2^-22)", 1.0 + 4 * u, 1.0 + 2 * u);
    // small accumulator, large product sum: is the ACCUMULATOR truncated during alignment?
    for (int k = 0; k < 16; ++k) { a[k] = 0.f; b[k] = 0.f; }
    a[0] = 1.f; b[0] = 1.f;
    run_case(a, b, 1.5f * u, "5. C=1.5*2^-24, product 1", 1.0 + 2 * u, 1.0);
    run_case(a, b, -0.25f * u, "5b. C=-2^-26, product 1", 1.0, 1.0 - u);
    // two products of opposite sign that cancel, plus a small one: internal sum exactness
    a[0] = 1.f; b[0] = 1.f; a[1] = 1.f; b[1] = -1.f; a[2] = 1.f; b[2] = ldexpf(1.f, -30);
    run_case(a, b, 0.f, "6. C=0, products 1 - 1 + 2^-30", ldexpf(1.f, -30), ldexpf(1.f, -30));
    a[1] = 0.f; b[1] = 0.f;
    run_case(a, b, -1.f, "6b. C=-1, products 1 + 2^-30", ldexpf(1.f, -30), ldexpf(1.f, -30));
  }
  // granule scan: C = -1 sets the largest exponent, product 1 cancels it, a third product -+2^-j survives exactly,
  // vanishes (truncation towards zero) or comes out as minus one granule (two's-complement floor)
  for (int sgn = -1; sgn <= 1; sgn += 2)
    for (int j = 22; j <= 34; j += 1) {
      std::vector<float> a(16, 0.f), b(16, 0.f);
      a[0] = 1.f; b[0] = 1.f; a[5] = 1.f; b[5] = sgn * ldexpf(1.f, -j);
      char name[96];
      snprintf(name, sizeof name, "7. C=-1, products 1 %c 2^-%d", sgn < 0 ? '-' : '+', j);
      const float got = run_case(a, b, -1.f, name, sgn * ldexp(1.0, -j), 0.0);
      if (got != 0.f) printf("      = %c2^%d\n", got < 0 ? '-' : '+', ilogbf(fabsf(got)));
    }
  // the same with the largest exponent set by a PRODUCT (C = 0, products 1 and -1)
  for (int sgn = -1; sgn <= 1; sgn += 2)
    for (int j = 22; j <= 34; j += 2) {
      std::vector<float> a(16, 0.f), b(16, 0.f);
      a[0] = 1.f; b[0] = 1.f; a[1] = 1.f; b[1] = -1.f; a[5] = 1.f; b[5] = sgn * ldexpf(1.f, -j);
      char name[96];
      snprintf(name, sizeof name, "8. C=0, products 1 - 1 %c 2^-%d", sgn < 0 ? '-' : '+', j);
      const float got = run_case(a, b, 0.f, name, sgn * ldexp(1.0, -j), 0.0);
      if (got != 0.f) printf("      = %c2^%d\n", got < 0 ? '-' : '+', ilogbf(fabsf(got)));
    }
  // small C against large products: C = -+2^-j, products 1 - 1
  for (int sgn = -1; sgn <= 1; sgn += 2)
    for (int j = 22; j <= 34; j += 2) {
      std::vector<float> a(16, 0.f), b(16, 0.f);
      a[0] = 1.f; b[0] = 1.f; a[1] = 1.f; b[1] = -1.f;
      char name[96];
      snprintf(name, sizeof name, "9. C=%c2^-%d, products 1 - 1", sgn < 0 ? '-' : '+', j);
      const float got = run_case(a, b, sgn * ldexpf(1.f, -j), name, sgn * ldexp(1.0, -j), 0.0);
      if (got != 0.f) printf("      = %c2^%d\n", got < 0 ? '-' : '+', ilogbf(fabsf(got)));
    }
  // all-positive statistics: operands |N(0,1)|, C = 0, chain of 64: the sums grow, ulp(result) is the unit, and a
  // floor-truncating alignment shows as a negative mean
  {
    srand(777);
    const int steps = 64, probs = 8192;
    const size_t n = (size_t)probs * steps * 16;
    std::vector<unsigned short> ha(n), hb(n);
    std::vector<float> hc(probs, 0.f);
    for (size_t i = 0; i < n; ++i) {
      ha[i] = bf16_bits(bf16_val(bf16_bits((float)fabs(nrand()))));
      hb[i] = bf16_bits(bf16_val(bf16_bits((float)fabs(nrand()))));
    }
    unsigned short *da, *db;
    float *dc, *dout;
    CK(hipMalloc(&da, n * 2)); CK(hipMalloc(&db, n * 2)); CK(hipMalloc(&dc, probs * 4)); CK(hipMalloc(&dout, probs * 4));
    CK(hipMemcpy(da, ha.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), probs * 4, hipMemcpyHostToDevice));
    chain_bf16_many<<<probs / 4, 256>>>(da, db, dc, steps, dout);
    std::vector<float> out(probs);
    CK(hipMemcpy(out.data(), dout, probs * 4, hipMemcpyDeviceToHost));
    double mean = 0, rms = 0;
    for (int p = 0; p < probs; ++p) {
      double exact = 0;
      for (int i = 0; i < steps * 16; ++i) exact += (double)bf16_val(ha[(size_t)p * steps * 16 + i]) * bf16_val(hb[(size_t)p * steps * 16 + i]);
      int e;
      frexp(exact, &e);
      const double err = ((double)out[p] - exact) / ldexp(1.0, e - 24);
      mean += err; rms += err * err;
    }
    printf("bf16 MFMA chain of 64, all-positive operands: error in ulp(result): mean %+.3f rms %.3f\n", mean / probs, sqrt(rms / probs));
    // the same products scaled by 2^-16 in ONE operand every other step (a stand-in for the small partial products of
    // a split GEMM riding on a large accumulator)
    for (size_t i = 0; i < n; ++i)
      if ((i / 16) % 2 == 1) ha[i] = bf16_bits(bf16_val(ha[i]) * ldexpf(1.f, -16));
    CK(hipMemcpy(da, ha.data(), n * 2, hipMemcpyHostToDevice));
    chain_bf16_many<<<probs / 4, 256>>>(da, db, dc, steps, dout);
    CK(hipMemcpy(out.data(), dout, probs * 4, hipMemcpyDeviceToHost));
    mean = rms = 0;
    for (int p = 0; p < probs; ++p) {
      double exact = 0;
      for (int i = 0; i < steps * 16; ++i) exact += (double)bf16_val(ha[(size_t)p * steps * 16 + i]) * bf16_val(hb[(size_t)p * steps * 16 + i]);
      int e;
      frexp(exact, &e);
      const double err = ((double)out[p] - exact) / ldexp(1.0, e - 24);
      mean += err; rms += err * err;
    }
    printf("   ... every other step scaled by 2^-16 (small products on a large accumulator): mean %+.3f rms %.3f\n", mean / probs, sqrt(rms / probs));
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(dout));
  }
  // statistics
  srand(12345);
  for (int pass = 0; pass < 2; ++pass) {
    const int steps = pass == 0 ? 1 : 64, probs = 8192;
    const size_t n = (size_t)probs * steps * 16;
    std::vector<unsigned short> ha(n), hb(n);
    std::vector<float> hc(probs);
    for (size_t i = 0; i < n; ++i) {
      ha[i] = bf16_bits(bf16_val(bf16_bits((float)nrand())));
      hb[i] = bf16_bits(bf16_val(bf16_bits((float)nrand())));
    }
    for (int p = 0; p < probs; ++p) hc[p] = pass == 0 ? (float)(nrand() * 8.0) : 0.f;
    unsigned short *da, *db;
    float *dc, *dout;
    CK(hipMalloc(&da, n * 2)); CK(hipMalloc(&db, n * 2)); CK(hipMalloc(&dc, probs * 4)); CK(hipMalloc(&dout, probs * 4));
    CK(hipMemcpy(da, ha.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), probs * 4, hipMemcpyHostToDevice));
    chain_bf16_many<<<probs / 4, 256>>>(da, db, dc, steps, dout);
    std::vector<float> out(probs);
    CK(hipMemcpy(out.data(), dout, probs * 4, hipMemcpyDeviceToHost));
    double mean_signed = 0, mean_tozero = 0, rms = 0;
    for (int p = 0; p < probs; ++p) {
      double exact = hc[p];
      for (int i = 0; i < steps * 16; ++i) exact += (double)bf16_val(ha[(size_t)p * steps * 16 + i]) * bf16_val(hb[(size_t)p * steps * 16 + i]);
      int e;
      frexp(fabs(exact) > 0 ? exact : 1.0, &e);
      const double ulp = ldexp(1.0, e - 24);
      const double err = ((double)out[p] - exact) / ulp;
      mean_signed += err;
      mean_tozero += err * (exact >= 0 ? 1.0 : -1.0);
      rms += err * err;
    }
    printf("bf16 MFMA chain of %2d (K = %4d), %d problems: error in ulp(result): mean signed %+.3f, mean towards +inf of |x| %+.3f, rms %.3f\n",
           steps, steps * 16, probs, mean_signed / probs, mean_tozero / probs, sqrt(rms / probs));
    // the f32 MFMA on the same products (operands widened: exact in fp32)
    const int fsteps = steps * 8;
    std::vector<float> fa((size_t)probs * fsteps * 2), fb(fa.size());
    for (size_t i = 0; i < fa.size(); ++i) { fa[i] = bf16_val(ha[i]); fb[i] = bf16_val(hb[i]); }
    float *dfa, *dfb;
    CK(hipMalloc(&dfa, fa.size() * 4)); CK(hipMalloc(&dfb, fa.size() * 4));
    CK(hipMemcpy(dfa, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfb, fb.data(), fa.size() * 4, hipMemcpyHostToDevice));
    chain_f32_many<<<probs / 4, 256>>>(dfa, dfb, dc, fsteps, dout);
    CK(hipMemcpy(out.data(), dout, probs * 4, hipMemcpyDeviceToHost));
    mean_signed = mean_tozero = rms = 0;
    for (int p = 0; p < probs; ++p) {
      double exact = hc[p];
      for (int i = 0; i < steps * 16; ++i) exact += (double)fa[(size_t)p * steps * 16 + i] * fb[(size_t)p * steps * 16 + i];
      int e;
      frexp(fabs(exact) > 0 ? exact : 1.0, &e);
      const double ulp = ldexp(1.0, e - 24);
      const double err = ((double)out[p] - exact) / ulp;
      mean_signed += err;
      mean_tozero += err * (exact >= 0 ? 1.0 : -1.0);
      rms += err * err;
    }
    printf("f32  MFMA chain of %3d (K = %4d), %d problems: error in ulp(result): mean signed %+.3f, mean towards +inf of |x| %+.3f, rms %.3f\n",
           fsteps, fsteps * 2, probs, mean_signed / probs, mean_tozero / probs, sqrt(rms / probs));
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(dout)); CK(hipFree(dfa)); CK(hipFree(dfb));
  }
  return 0;
}
