// Diagnostic microbenchmark: cost of LDS read-modify-write flavours per wave-instruction (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, float* __restrict__ out, int iters) {
  __shared__ float buf[4096];
  __shared__ unsigned ubuf[4096];
  __shared__ double dbuf[2048];
  __shared__ unsigned long long lbuf[2048];
  for (int i = threadIdx.x; i < 4096; i += 256) { buf[i] = 0.f; ubuf[i] = 0u; }
  for (int i = threadIdx.x; i < 2048; i += 256) { dbuf[i] = 0.0; lbuf[i] = 0ull; }
  __syncthreads();
  const int base = idx[blockIdx.x * 256 + threadIdx.x];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int a = (base + it * 67) & 4095;
    if (MODE == 0) atomicAdd(&buf[a], 1.0f);                       // ds_add_f32
    else if (MODE == 1) atomicAdd(&ubuf[a], 1u);                   // ds_add_u32
    else if (MODE == 2) acc += atomicAdd(&buf[a], 1.0f);           // ds_add_rtn_f32
    else if (MODE == 3) { float v = buf[a]; buf[a] = v + 1.0f; }   // plain RMW (racy; cost reference)
    else if (MODE == 4) acc += buf[a];                             // read only
    else if (MODE == 5) atomicMax(&ubuf[a], (unsigned)it);         // ds_max_u32
    else if (MODE == 6) atomicAdd(&dbuf[a & 2047], 1.0);           // ds_add_f64
    else if (MODE == 7) atomicAdd(&lbuf[a & 2047], 1ull);          // ds_add_u64
  }
  __syncthreads();
  float s = acc;
  for (int i = threadIdx.x; i < 4096; i += 256) s += buf[i] + (float)ubuf[i];
  for (int i = threadIdx.x; i < 2048; i += 256) s += (float)dbuf[i] + (float)lbuf[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks = 256 * 8, iters = 512;
  std::vector<int> h(blocks * 256);
  int *d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4);
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_f32", "plain_rmw", "read_only", "ds_max_u32",
                         "ds_add_f64", "ds_add_u64"};
  for (int pat = 0; pat < 3; ++pat) {
    for (size_t i = 0; i < h.size(); ++i) {
      int lane = i & 255;
      h[i] = pat == 0 ? lane : (pat == 1 ? (rand() & 4095) : (lane / 2));   // consecutive / random / pairs collide
    }
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 8; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: k<0><<<blocks, 256>>>(d, o, iters); break;
          case 1: k<1><<<blocks, 256>>>(d, o, iters); break;
          case 2: k<2><<<blocks, 256>>>(d, o, iters); break;
          case 3: k<3><<<blocks, 256>>>(d, o, iters); break;
          case 4: k<4><<<blocks, 256>>>(d, o, iters); break;
          case 5: k<5><<<blocks, 256>>>(d, o, iters); break;
          case 6: k<6><<<blocks, 256>>>(d, o, iters); break;
          default: k<7><<<blocks, 256>>>(d, o, iters); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // wave-instructions per CU: blocks*4 waves*iters / 256 CUs
      double wi_per_cu = (double)blocks * 4 * iters / 256.0;
      printf("pattern %d %-16s %8.3f ms  ~%6.1f cycles/wave-instr/CU (at 2.4GHz)\n", pat, names[mode], ms,
             ms * 1e-3 * 2.4e9 / wi_per_cu);
    }
  }
  return 0;
}
