// Diagnostic: would the v_mfma_f32_16x16x32_bf16 shape pay in the bf16x3 GEMM's k-loop?  
// The shape consumes K = 32 per instruction, so the operand images of a pipeline stage double: at the shipped 128 x 256
// workgroup tile and two stages per operand that is 144 KiB of LDS - ONE 8-wave workgroup per CU instead of two.  This
// bench runs the k-loop skeleton of both candidates on random bf16 data resident in LDS - fragment reads
// (ds_read_b128), the six-product MFMA block, three (six) 16-byte LDS stores per lane standing in for the split store,
// one raw barrier per k-tile - WITHOUT global memory traffic and without the split arithmetic, i.e. the best case of
// either structure:
//   A  32x32x16, k-tile 16: 12 fragment reads, 24 MFMAs per wave and tile; 72 KiB per workgroup, TWO workgroups per CU
//   B  16x16x32, k-tile 32: 24 fragment reads, 96 MFMAs per wave and tile; 144 KiB per workgroup, ONE workgroup per CU
//   C  16x16x32 as B but (hypothetically) two workgroups per CU - the stages overlapped in 64 KiB, possible only because
//      the data do not matter here: what the shape would be worth if the LDS were free
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_kloop_bench.hip -o build/tools/mfma_kloop_bench
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
__device__ __forceinline__ u32x4 rnd_chunk(unsigned seed) {      // eight bf16 in (-2, 2), random significands
  u32x4 w;
  for (int i = 0; i < 4; ++i) {
    const unsigned h = hash(seed * 4 + i);
    const unsigned lo = (h & 0x807fu) | 0x3f00u | ((h >> 9) & 0x80u), hi = ((h >> 16) & 0x807fu) | 0x3f00u;
    w[i] = lo | (hi << 16);
  }
  return w;
}

// LDS image of one operand stage: [plane 3][k-group KG][row 128] chunks of 16 B (KG = 2 for k-tile 16, 4 for 32)
template <int SHAPE>
__global__ void __launch_bounds__(512) kloop(float* out, int tiles, int lds_chunks, int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4* img = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = wave >> 2, lw = wave & 3, wm = lw >> 1, wn = lw & 1;
  for (int i = tid; i < lds_chunks; i += 512) img[i] = rnd_chunk(blockIdx.x * 65536 + i);
  __syncthreads();
  float s = 0.f;
  if (SHAPE == 32) {
    constexpr int SIMG = 3 * 2 * 128;                 // chunks per operand stage
    // [2 subs][2 B stages] | [2 A stages]
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int t = 0; t < tiles; ++t) {
      const int cur = t & 1;
      const u32x4* As = img + (4 + cur) * stride + lh * 128 + wm * 64 + li;
      const u32x4* Bs = img + (sub * 2 + cur) * stride + lh * 128 + wn * 64 + li;
      u32x4 a[3][2], b[2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { a[pl][0] = As[pl * 256]; a[pl][1] = As[pl * 256 + 32]; }
#pragma unroll
      for (int pb = 2; pb >= 0; --pb) {
        b[0] = Bs[pb * 256]; b[1] = Bs[pb * 256 + 32];
#pragma unroll
        for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[pa][tm]),
                                                                    __builtin_bit_cast(bf16x8, b[tn]), acc[tm][tn], 0, 0, 0);
      }
      // the split store of the next tile: three chunks per lane into the other B stage
      u32x4* st = img + (sub * 2 + (cur ^ 1)) * stride + (tid & 255);
      const u32x4 v = a[0][0];
      st[0] = v; st[256] = v; st[512] = v;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    constexpr int SIMG = 3 * 4 * 128;
    const int li = lane & 15, lg = lane >> 4;           // row within a 16-row block, k-group of 8
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int t = 0; t < tiles; ++t) {
      const int cur = t & 1;
      const u32x4* As = img + (4 + cur) * stride + lg * 128 + wm * 64 + li;
      const u32x4* Bs = img + (sub * 2 + cur) * stride + lg * 128 + wn * 64 + li;
      u32x4 a[3][4], b[4];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[pl][i] = As[pl * 512 + 16 * i];
#pragma unroll
      for (int pb = 2; pb >= 0; --pb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[pb * 512 + 16 * j];
#pragma unroll
        for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[pa][tm]),
                                                                    __builtin_bit_cast(bf16x8, b[tn]), acc[tm][tn], 0, 0, 0);
      }
      u32x4* st = img + (sub * 2 + (cur ^ 1)) * stride + (tid & 255);
      const u32x4 v = a[0][0];
#pragma unroll
      for (int q = 0; q < 6; ++q) st[q * 256] = v;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE>
void run(float* out, const char* name, size_t lds_bytes, int wg_per_cu, int stride = 0) {
  const int kt = SHAPE == 32 ? 16 : 32;
  const int tiles = 8192 * 16 / kt, grid = 256 * wg_per_cu;
  const int simg = SHAPE == 32 ? 768 : 1536;
  if (stride == 0) stride = simg;
  const int chunks = 5 * stride + simg;      // (stride < simg: the stages overlap - a throughput model, the data do not matter)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kloop<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float last = 0.f, best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kloop<SHAPE>, dim3(grid), dim3(512), lds_bytes, 0, out, tiles, chunks, stride);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&last, e0, e1);
    if (last < best) best = last;
  }
  // executed bf16 MFMA flops: six products of a 64 x 64 x kt wave tile, eight waves per workgroup
  const double flops = (double)grid * 8 * tiles * 6.0 * 2.0 * 64 * 64 * kt;
  printf("%-64s last %8.3f ms = %7.1f TF executed (%6.1f TF fp32-equivalent)   best %7.1f TF\n", name, last,
         flops / last / 1e9, flops / last / 1e9 / 6, flops / best / 1e9);
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 2 * 512 * sizeof(float));
  for (int round = 0; round < 2; ++round) {
    run<32>(out, "A  32x32x16, k-tile 16, 72 KiB, two workgroups per CU", 76 * 1024, 2);
    run<16>(out, "B  16x16x32, k-tile 32, 144 KiB, ONE workgroup per CU", 150 * 1024, 1);
    run<16>(out, "C  16x16x32, k-tile 32, two workgroups per CU (stages overlapped in 64 KiB)", 66 * 1024, 2, 512);
  }
  return 0;
}
