// Diagnostic: sustained bf16 MFMA rate of the two shapes on gfx950, registers only, random vs zero operands.
// A wave owns a 64x64 accumulator tile and runs one k = 32 step per iteration: 8 v_mfma_f32_32x32x16_bf16 (2x2 blocks,
// two k halves) or 16 v_mfma_f32_16x16x32_bf16 (4x4 blocks).  512-thread workgroups, two per CU (4 waves per SIMD, the
// occupancy of the split GEMM kernels).  Prints TF of executed bf16 MFMA work.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o build/tools/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// eight bf16 values in (-2, 2) with random significands (or zeros)
__device__ __forceinline__ bf16x8 frag(unsigned seed, bool zeros) {
  u32x4 w;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = hash(seed * 4 + i);
    // two bf16: sign | exponent 0x3f (0.5..2 scaled) | 7 random mantissa bits
    const unsigned lo = (h & 0x807fu) | 0x3f00u | ((h >> 9) & 0x80u), hi = ((h >> 16) & 0x807fu) | 0x3f00u;
    w[i] = zeros ? 0u : (lo | (hi << 16));
  }
  return __builtin_bit_cast(bf16x8, w);
}

template <int SHAPE>
__global__ void __launch_bounds__(512) kern(float* out, int iters, int zeros) {
  extern __shared__ float lds[];
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  float s = 0.f;
  if (SHAPE == 32) {
    bf16x8 a[2][2], b[2][2];   // [k half][block]
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 2; ++i) { a[k][i] = frag(t * 16 + k * 2 + i, zeros); b[k][i] = frag(t * 16 + 8 + k * 2 + i, zeros); }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag(t * 16 + i, zeros); b[i] = frag(t * 16 + 8 + i, zeros); }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  out[t] = s;
}

template <int SHAPE>
void run(float* out, int zeros) {
  const int iters = 20000, grid = 256 * 2;
  const size_t lds = 76 * 1024;   // two workgroups per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&kern<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f, last = 0.f;
  for (int rep = 0; rep < 6; ++rep) {   // back to back: the later repetitions run at the sustained clock
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<SHAPE>, dim3(grid), dim3(512), lds, 0, out, iters, zeros);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&last, e0, e1);
    if (last < best) best = last;
  }
  const double flops = (double)grid * 8 * iters * 2.0 * 64 * 64 * 32;
  printf("%-22s %-7s last %7.3f ms = %7.1f TF   best %7.3f ms = %7.1f TF\n",
         SHAPE == 32 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", zeros ? "zeros" : "random", last,
         flops / last / 1e9, best, flops / best / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 256 * 2 * 512 * sizeof(float));
  for (int round = 0; round < 2; ++round)
    for (int z = 0; z < 2; ++z) { run<32>(out, z); run<16>(out, z); }
  return 0;
}
