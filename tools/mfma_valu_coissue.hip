// Diagnostic: can packed-FP32 VALU work ride along an FP32 MFMA stream on gfx950?
// Each wave runs a dependency-free stream of v_mfma_f32_32x32x2_f32 with NV v_pk_fma_f32 (2 FMA per lane)
// interleaved per MFMA, registers only.  Prints MFMA TF, VALU TF and their sum for NV = 0..16.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>
__global__ void __launch_bounds__(256) kern(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x2 v[16];
  for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x, 1.0f + i};
  const f32x2 m = {1.0001f, 0.9999f}, a = {0.5f, 0.25f};
  const float x = 1.0f + threadIdx.x, y = 2.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[k], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(k * NV + j) & 15] = __builtin_elementwise_fma(v[(k * NV + j) & 15], m, a);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float* out, int wg_per_cu) {
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // dynamic LDS caps the occupancy at wg_per_cu workgroups per CU (an uncapped grid of 4 per CU is
  // packed unevenly by the dispatcher and reads 123 TF)
  const size_t lds = ((size_t)(160 * 1024 / wg_per_cu) - 2048) & ~(size_t)1023;   // margin for the allocation granularity
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern<NV>, dim3(grid), dim3(256), lds, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern<NV>, dim3(grid), dim3(256), lds, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * 4, mf = waves * iters * 4 * 4096.0, vf = waves * iters * 4.0 * NV * 64 * 4;
  printf("wg/cu %d  pk_fma per MFMA %2d : MFMA %6.1f TF  VALU %6.1f TF  sum %6.1f TF  (%.3f ms)\n", wg_per_cu, NV,
         mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, ms);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 3, 4}) {
    run<0>(out, w);
    if (w == 3) continue;
    run<2>(out, w); run<4>(out, w); run<6>(out, w); run<8>(out, w); run<12>(out, w); run<16>(out, w);
  }
  return 0;
}
