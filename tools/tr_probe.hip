// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  LDS tile [16 rows][64 cols] of u16, value = row * 256 + col,
// row stride 128 B.  Lane 4q+p of each 16-lane group g supplies the address of (row r0(g) + q, cols c0(g) + 4p .. +3);
// prints, per lane, the four 16-bit values it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = (uint16_t)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r0 = 4 * (g >> 1), c0 = 16 * (g & 1);     // groups 0,1: rows 0-3, cols 0-15 / 16-31; groups 2,3: rows 4-7
  const uint32_t addr = (uint32_t)(uintptr_t)tile + (uint32_t)(((r0 + q) * 64 + c0 + 4 * p) * 2);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)(v >> (16 * e));
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (r%d,c%2d)", h[l * 4 + e] >> 8, h[l * 4 + e] & 255);
    printf("\n");
  }
  return 0;
}
