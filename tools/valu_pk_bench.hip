// Diagnostic microbenchmark: issue cost of fp32 vector instructions per 64-wide wave on gfx950 - plain, packed
// (v_pk_*_f32: two fp32 operations per lane), double, and the conversions / special functions the advection kernels
// use.  Eight independent accumulator chains per wave, 8 waves per SIMD: throughput, not latency.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_pk_bench.hip -o build/tools/valu_pk_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, int iters, float seed) {
  float a[8]; f2 p[8]; double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; d[i] = a[i]; }
  const float m = 0.999f, c = 1e-3f;
  const f2 m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
        else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        else if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)m), "v"((double)c));
        else if (MODE == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        else if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        else if (MODE == 7) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        else if (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        else if (MODE == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(3));
        else if (MODE == 10) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        else if (MODE == 11) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(a[i]) : "v"(m));
        else if (MODE == 12) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks = 256 * 8, iters = 2000;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  float* o; hipMalloc(&o, blocks * 256 * 4);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f64", "v_cvt_f64_f32",
                         "v_rcp_f32", "v_fract_f32", "v_mul_f32", "v_add_u32", "v_fmac_f32_e32", "v_fmaak_f32",
                         "v_mul_f32_e64"};
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const double ghz = pr.clockRate * 1e-6;
  printf("clockRate %.2f GHz (nominal; cycles below assume it)\n", ghz);
  for (int mode = 0; mode < 13; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: k<0><<<blocks, 256>>>(o, iters, 1.f); break;
        case 1: k<1><<<blocks, 256>>>(o, iters, 1.f); break;
        case 2: k<2><<<blocks, 256>>>(o, iters, 1.f); break;
        case 3: k<3><<<blocks, 256>>>(o, iters, 1.f); break;
        case 4: k<4><<<blocks, 256>>>(o, iters, 1.f); break;
        case 5: k<5><<<blocks, 256>>>(o, iters, 1.f); break;
        case 6: k<6><<<blocks, 256>>>(o, iters, 1.f); break;
        case 7: k<7><<<blocks, 256>>>(o, iters, 1.f); break;
        case 8: k<8><<<blocks, 256>>>(o, iters, 1.f); break;
        case 9: k<9><<<blocks, 256>>>(o, iters, 1.f); break;
        case 10: k<10><<<blocks, 256>>>(o, iters, 1.f); break;
        case 11: k<11><<<blocks, 256>>>(o, iters, 1.f); break;
        default: k<12><<<blocks, 256>>>(o, iters, 1.f); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // wave-instructions per SIMD: 8 waves * iters * 32
    const double winstr = 8.0 * iters * 32;
    printf("%-14s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[mode], ms, ms * 1e-3 * ghz * 1e9 / winstr);
  }
  return 0;
}
