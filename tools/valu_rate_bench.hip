// Diagnostic microbenchmark: VALU issue rate per wave-instruction on gfx950 for the instruction
// classes of the advection kernels (plain fp32 FMA, packed fp32, transcendental, f64 helpers, DPP).
// Answers: does v_pk_fma_f32 retire two FMAs per lane at the v_fma_f32 issue cost?  (It does not help
// if a packed instruction costs twice the issue cycles.)
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define OP8_2(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
  INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
#define OP8_1(INS) asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" \
  INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
#define OP8_3(INS) asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
  INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))

template <int MODE>
__global__ void __launch_bounds__(256) k(float* __restrict__ out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.001f;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const v2f pm = {m, m}, pc = {c, c};
  double d0 = a0, d1 = a1;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) {          // 8 independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                     "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                     "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c));
      } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 (= 8 FMAs per lane)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n"
                     "v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
      } else if (MODE == 2) {   // 4 v_pk_mul_f32 + 4 v_pk_add_f32
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n"
                     "v_pk_add_f32 %3, %3, %5\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n"
                     "v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
      } else if (MODE == 3) {   // 8 v_rcp_f32 (transcendental unit)
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                     "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 4) {   // 4 v_cvt_f64_f32 + 4 v_add_f64 (fixed-point conversion of the scatter)
        asm volatile("v_cvt_f64_f32 %0, %2\n v_add_f64 %0, %0, %1\n v_cvt_f64_f32 %1, %3\n v_add_f64 %1, %1, %0\n"
                     "v_cvt_f64_f32 %0, %4\n v_add_f64 %0, %0, %1\n v_cvt_f64_f32 %1, %5\n v_add_f64 %1, %1, %0\n"
                     : "+v"(d0), "+v"(d1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      } else if (MODE == 5) {   // 8 v_add_f32 with DPP row_shr:1
        asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %2, %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %4, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %5, %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %6, %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %7, %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 6) {   // 8 v_cndmask / v_min / v_max mix (select-heavy code)
        asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                     "v_max_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 10) { OP8_2("v_mul_f32");
      } else if (MODE == 11) { OP8_2("v_add_f32");
      } else if (MODE == 12) { OP8_2("v_sub_f32");
      } else if (MODE == 13) { OP8_2("v_fmac_f32");
      } else if (MODE == 14) { OP8_1("v_floor_f32");
      } else if (MODE == 15) { OP8_1("v_cvt_i32_f32");
      } else if (MODE == 16) { OP8_1("v_cvt_f32_i32");
      } else if (MODE == 17) { OP8_2("v_add_u32");
      } else if (MODE == 18) { OP8_2("v_mul_lo_u32");
      } else if (MODE == 19) { OP8_2("v_lshlrev_b32");
      } else if (MODE == 20) { OP8_2("v_and_b32");
      } else if (MODE == 21) { OP8_3("v_med3_f32");
      } else if (MODE == 22) { OP8_3("v_bfi_b32");
      } else if (MODE == 23) { OP8_3("v_mad_u32_u24");
      } else if (MODE == 24) { OP8_2("v_cndmask_b32");   // vcc implicit
      } else if (MODE == 25) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                     "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n"
                     :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
      } else if (MODE == 26) { OP8_1("v_fract_f32");
      } else if (MODE == 27) { OP8_1("v_rndne_f32");
      } else if (MODE == 28) { OP8_1("v_mov_b32");
      } else if (MODE == 29) { OP8_3("v_add3_u32");
      } else if (MODE == 30) { OP8_3("v_lshl_add_u32");
      } else if (MODE == 31) { OP8_1("v_sin_f32");
      } else if (MODE == 32) { OP8_2("v_max_i32");
      } else if (MODE == 33) { OP8_1("v_rsq_f32");
      } else if (MODE == 7) {   // 8 v_sqrt_f32
        asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                     "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y +
                                        p2.x + p2.y + p3.x + p3.y + (float)d0 + (float)d1;
}

template <int MODE>
static void run(const char* name, float* o, int waves_per_simd, double lane_ops_per_instr) {
  const int blocks = 256 * waves_per_simd, iters = 4096;   // 256 threads = 1 wave per SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(o, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double instr_per_wave = (MODE == 1 ? 4.0 : 8.0) * 8.0 * iters;
  const double wave_instr_per_simd = instr_per_wave * waves_per_simd;
  const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
  printf("%-34s %d waves/SIMD  %7.3f ms  %5.2f cyc/wave-instr/SIMD (at 2.4 GHz)  %6.1f Glane-op/s/SIMD\n", name,
         waves_per_simd, ms, cyc, lane_ops_per_instr * 64.0 * wave_instr_per_simd / (ms * 1e-3) / 1e9);
}

int main() {
  float* o; hipMalloc(&o, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", o, w, 1);
    run<1>("v_pk_fma_f32 (2 FMA/lane)", o, w, 2);
    run<2>("v_pk_mul_f32/v_pk_add_f32", o, w, 2);
    run<3>("v_rcp_f32", o, w, 1);
    run<7>("v_sqrt_f32", o, w, 1);
    run<4>("v_cvt_f64_f32 + v_add_f64", o, w, 1);
    run<5>("v_add_f32 dpp row_shr:1", o, w, 1);
    run<6>("v_min/v_max_f32", o, w, 1);
  }
  const char* names[] = {"v_mul_f32","v_add_f32","v_sub_f32","v_fmac_f32","v_floor_f32","v_cvt_i32_f32","v_cvt_f32_i32",
    "v_add_u32","v_mul_lo_u32","v_lshlrev_b32","v_and_b32","v_med3_f32","v_bfi_b32","v_mad_u32_u24","v_cndmask_b32",
    "v_cmp_lt_f32","v_fract_f32","v_rndne_f32","v_mov_b32","v_add3_u32","v_lshl_add_u32","v_sin_f32","v_max_i32","v_rsq_f32"};
  {
    const int w = 8;
    run<10>(names[0], o, w, 1); run<11>(names[1], o, w, 1); run<12>(names[2], o, w, 1); run<13>(names[3], o, w, 1);
    run<14>(names[4], o, w, 1); run<15>(names[5], o, w, 1); run<16>(names[6], o, w, 1); run<17>(names[7], o, w, 1);
    run<18>(names[8], o, w, 1); run<19>(names[9], o, w, 1); run<20>(names[10], o, w, 1); run<21>(names[11], o, w, 1);
    run<22>(names[12], o, w, 1); run<23>(names[13], o, w, 1); run<24>(names[14], o, w, 1); run<25>(names[15], o, w, 1);
    run<26>(names[16], o, w, 1); run<27>(names[17], o, w, 1); run<28>(names[18], o, w, 1); run<29>(names[19], o, w, 1);
    run<30>(names[20], o, w, 1); run<31>(names[21], o, w, 1); run<32>(names[22], o, w, 1); run<33>(names[23], o, w, 1);
  }
  return 0;
}
