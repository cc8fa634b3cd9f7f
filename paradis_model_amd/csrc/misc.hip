// a9 GlobalBias map (reference model/blocks.py:188-196) and the elementwise / reduction glue of
// the ADR layer step (reference model/paradis.py:239-253): activation, gated blend, bias grads.
// All HBM-bound streaming kernels (float4 where alignment allows).
#include <algorithm>
#include "common.h"

namespace {

inline int stream_blocks(int64_t n_items) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((n_items + 255) / 256, 256 * 16));
}

// ------------------------------------------------------------------ global bias map, forward
// grid (ceil(W/64), H, Cin): m8[c,h,w] = sum_r (A[c,r] U[r,h]) V[r,w].  64 columns per workgroup, the rank range in
// four quarters (one per wave) summed in quarter order: at W = 64 a 256-column workgroup had three idle waves and
// walked all R ranks in one dependent chain (23 us for 2 M multiply-adds).
__global__ void __launch_bounds__(256)
gbias_m8_kernel(const float* __restrict__ A, const float* __restrict__ U, const float* __restrict__ V,
                float* __restrict__ m8, int Cin, int R, int H, int W) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int w = min(blockIdx.x * 64 + tx, W - 1), h = blockIdx.y, c = blockIdx.z;
  const int rq = (R + 3) / 4, r1 = min(R, (q + 1) * rq);
  // rank loop in batches of independent loads (one dependent load per iteration made this 50 us)
  float acc = 0.f;
  int r = q * rq;
  for (; r + 8 <= r1; r += 8) {
    float a[8], u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = A[c * R + r + j]; u[j] = U[(r + j) * H + h]; v[j] = V[(r + j) * W + w]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] * u[j] * v[j];
  }
  for (; r < r1; ++r) acc += A[c * R + r] * U[r * H + h] * V[r * W + w];
  part[q][tx] = acc;
  __syncthreads();
  if (q == 0 && blockIdx.x * 64 + tx < W)
    m8[((int64_t)c * H + h) * W + w] = ((part[0][tx] + part[1][tx]) + part[2][tx]) + part[3][tx];
}

// Large grids: ROWS latitude rows per workgroup.  The rank-dependent factors a[c,r] u[r,h] of the workgroup's rows are
// formed once into LDS (wave-uniform broadcast reads afterwards), so the rank loop is one coalesced V load per rank
// feeding ROWS independent accumulators, eight ranks in flight.  (One row per workgroup, every factor a scalar load
// in front of its use: 398 us for the 33 MB map of 721x1440.)  Per output the same products (a u) v in the same
// order as the kernel above.  R <= GM8_MAXR.
constexpr int GM8_MAXR = 256;
template <int ROWS>
__global__ void __launch_bounds__(256)
gbias_m8_rows_kernel(const float* __restrict__ A, const float* __restrict__ U, const float* __restrict__ V,
                     float* __restrict__ m8, int Cin, int R, int H, int W) {
  __shared__ float part[4][ROWS][64];
  __shared__ float coef[GM8_MAXR][ROWS];
  const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int w = min(blockIdx.x * 64 + tx, W - 1), h0 = blockIdx.y * ROWS, c = blockIdx.z;
  for (int i = threadIdx.x; i < R * ROWS; i += 256) {
    const int r = i / ROWS, j = i - r * ROWS;
    coef[r][j] = A[c * R + r] * U[r * H + min(h0 + j, H - 1)];
  }
  __syncthreads();
  const int rq = (R + 3) / 4, r1 = min(R, (q + 1) * rq);
  float acc[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) acc[j] = 0.f;
  int r = q * rq;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = V[(r + k) * W + w];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < ROWS; ++j) acc[j] += coef[r + k][j] * v[k];
  }
  for (; r < r1; ++r) {
    const float v = V[r * W + w];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) acc[j] += coef[r][j] * v;
  }
#pragma unroll
  for (int j = 0; j < ROWS; ++j) part[q][j][tx] = acc[j];
  __syncthreads();
  if (q == 0 && blockIdx.x * 64 + tx < W) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j)
      if (h0 + j < H)
        m8[((int64_t)c * H + h0 + j) * W + w] = ((part[0][j][tx] + part[1][j][tx]) + part[2][j][tx]) + part[3][j][tx];
  }
}

// out[o,p] = sum_c Wm[o*ldo + c*ldc] * in[c,p]      (projection and its transpose)
// grid (ceil(P/256), Cout)
__global__ void __launch_bounds__(256)
small_mix_kernel(const float* __restrict__ Wm, int ldo, int ldc, const float* __restrict__ in,
                 float* __restrict__ out, int Cout, int Cin, int64_t P) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  if (p >= P) return;
  float acc = 0.f;
  for (int c = 0; c < Cin; ++c) acc += Wm[(int64_t)o * ldo + (int64_t)c * ldc] * in[(int64_t)c * P + p];
  out[(int64_t)o * P + p] = acc;
}

// gm8[c,p] = sum_o Pw[o,c] * gmap[o,p]   (c < CIN <= 16 kept in registers)
// One workgroup = 64 pixels x 16 slices of the long sum over Co (one wave per slice: the row of Pw is wave-uniform);
// the slices meet in LDS and are added in slice order - no atomics, no zero fill, the same bits on every run.
// (Rounds 1-3 split Co over the grid and combined the chunks with float atomicAdd: run-to-run differences in the
// last bit of the GlobalBias gradients, which AdamW's normalisation can turn into visible parameter differences.)
constexpr int GM8_SLICES = 16;
#ifndef GM8_BATCH                // (A/B builds)
#define GM8_BATCH 16
#endif
template <int CIN>
__global__ void __launch_bounds__(64 * GM8_SLICES)
gbias_gm8_kernel(const float* __restrict__ Pw, const float* __restrict__ gmap, float* __restrict__ gm8,
                 int Cin, int Co, int64_t P) {
  __shared__ float part[GM8_SLICES][CIN][64];
  const int px = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p = (int64_t)blockIdx.x * 64 + px;
  const int64_t pc = min(p, P - 1);
  const int per = (Co + GM8_SLICES - 1) / GM8_SLICES, o0 = slice * per, o1 = min(o0 + per, Co);
  float acc[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
  int o = o0;
  for (; o + GM8_BATCH <= o1; o += GM8_BATCH) {      // sixteen independent row loads in flight (round 5; eight before: the
    float g[GM8_BATCH];                              //  wave's 64-row walk is a chain of batches, each a memory round trip)
#pragma unroll
    for (int j = 0; j < GM8_BATCH; ++j) g[j] = gmap[(int64_t)(o + j) * P + pc];
#pragma unroll
    for (int j = 0; j < GM8_BATCH; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        if (c < Cin) acc[c] += Pw[(int64_t)(o + j) * Cin + c] * g[j];
  }
  for (; o + 8 <= o1; o += 8) {
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = gmap[(int64_t)(o + j) * P + pc];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        if (c < Cin) acc[c] += Pw[(int64_t)(o + j) * Cin + c] * g[j];
  }
  for (; o < o1; ++o) {
    const float g = gmap[(int64_t)o * P + pc];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
      if (c < Cin) acc[c] += Pw[(int64_t)o * Cin + c] * g;
  }
#pragma unroll
  for (int c = 0; c < CIN; ++c) part[slice][c][px] = acc[c];
  __syncthreads();
  if (slice < Cin && p < P) {          // wave `slice` finishes channel c = slice
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < GM8_SLICES; ++k) s += part[k][slice][px];
    gm8[(int64_t)slice * P + p] = s;
  }
}

// ------------------------------------------------------------------ global bias map, backward
// gPw[o,c] = sum_p gmap[o,p] * m8[c,p]; one workgroup per (o,c).  (A one-workgroup-per-o form that reads the row of
// gmap once - 109 instead of 809 MB at 128x256 - was measured in round 4: 383 us against 180 us, 32 against 15 us at
// 32x64: 1024 workgroups walking 128 dependent iterations lose more than the re-reads through L2 cost.)
__global__ void __launch_bounds__(256)
gbias_gpw_kernel(const float* __restrict__ gmap, const float* __restrict__ m8,
                 float* __restrict__ gPw, int Cin, int64_t P) {
  __shared__ float red[4];
  const int o = blockIdx.x / Cin, c = blockIdx.x - o * Cin;
  float acc = 0.f;
  for (int64_t p = threadIdx.x; p < P; p += 256) acc += gmap[(int64_t)o * P + p] * m8[(int64_t)c * P + p];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) gPw[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// The same sums with RW output rows per workgroup and 16-byte loads (round 5; the larger grids): a workgroup walks the
// whole pixel range once - its RW rows of gmap come from HBM once, the Cin rows of m8 from L2 (Co / RW times 4 Cin P
// bytes; the (o, c)-workgroup form above pulls every row of gmap Cin times: 809 MB at 128x256, Co = 1024).  RW + Cin
// float4 loads in flight per thread and iteration.  Needs P % 4 == 0 and 16-byte aligned rows.
template <int RW, int CIN>
__global__ void __launch_bounds__(256)
gbias_gpw_rows_kernel(const float* __restrict__ gmap, const float* __restrict__ m8, float* __restrict__ gPw, int Cin,
                      int Co, int64_t P4) {
  __shared__ float red[4][RW * CIN];
  const int o0 = blockIdx.x * RW;
  float acc[RW][CIN];
#pragma unroll
  for (int j = 0; j < RW; ++j)
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[j][c] = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(gmap);
  const float4* m4 = reinterpret_cast<const float4*>(m8);
  for (int64_t q = threadIdx.x; q < P4; q += 256) {
    float4 m[CIN], g[RW];
#pragma unroll
    for (int c = 0; c < CIN; ++c) m[c] = m4[(int64_t)min(c, Cin - 1) * P4 + q];
#pragma unroll
    for (int j = 0; j < RW; ++j) g[j] = g4[(int64_t)min(o0 + j, Co - 1) * P4 + q];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        acc[j][c] += (g[j].x * m[c].x + g[j].y * m[c].y) + (g[j].z * m[c].z + g[j].w * m[c].w);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < RW; ++j)
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float sum = wave_sum_dpp(acc[j][c]);
      if (lane == 0) red[wave][j * CIN + c] = sum;
    }
  __syncthreads();
  if (threadIdx.x < RW * CIN) {
    const int j = threadIdx.x / CIN, c = threadIdx.x - j * CIN;
    if (o0 + j < Co && c < Cin)
      gPw[(int64_t)(o0 + j) * Cin + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// T1[c,r,h] = sum_w gm8[c,h,w] V[r,w]   (one wave per output)
__global__ void __launch_bounds__(256)
gbias_t1_kernel(const float* __restrict__ gm8, const float* __restrict__ V, float* __restrict__ T1,
                int Cin, int R, int H, int W) {
  const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (wid >= (int64_t)Cin * R * H) return;
  const int h = (int)(wid % H), r = (int)((wid / H) % R), c = (int)(wid / ((int64_t)H * R));
  float acc = 0.f;
  for (int w = threadIdx.x & 63; w < W; w += 64) acc += gm8[((int64_t)c * H + h) * W + w] * V[r * W + w];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) T1[wid] = acc;
}

// T2[c,r,w] = sum_h gm8[c,h,w] U[r,h]
__global__ void __launch_bounds__(256)
gbias_t2_kernel(const float* __restrict__ gm8, const float* __restrict__ U, float* __restrict__ T2,
                int Cin, int R, int H, int W) {
  const int64_t total = (int64_t)Cin * R * W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int w = (int)(idx % W), r = (int)((idx / W) % R), c = (int)(idx / ((int64_t)W * R));
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc += gm8[((int64_t)c * H + h) * W + w] * U[r * H + h];
    T2[idx] = acc;
  }
}

// gA[c,r] = sum_h T1[c,r,h] U[r,h];  gU[r,h] = sum_c T1[c,r,h] A[c,r];  gV[r,w] = sum_c T2[c,r,w] A[c,r]
// (gA: one WAVE per output, lanes along h - a thread per output walked H dependent iterations of a strided row and set
//  the kernel's duration: 33 us at 128x256; gU, gV: a thread per output, coalesced along h / w)
__global__ void __launch_bounds__(256)
gbias_finish_kernel(const float* __restrict__ T1, const float* __restrict__ T2,
                    const float* __restrict__ A, const float* __restrict__ U,
                    float* __restrict__ gA, float* __restrict__ gU, float* __restrict__ gV, int Cin,
                    int R, int H, int W) {
  const int nA = Cin * R, nU = R * H, nV = R * W;
  const int wavesA = (nA + 3) / 4;               // workgroups that hold the gA waves (four per workgroup)
  if ((int)blockIdx.x < wavesA) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= nA) return;
    const int c = idx / R, r = idx - c * R;
    float acc = 0.f;
    for (int h = lane; h < H; h += 64) acc += T1[((int64_t)c * R + r) * H + h] * U[r * H + h];
    acc = wave_sum_dpp(acc);
    if (lane == 0) gA[idx] = acc;
    return;
  }
  const int idx = ((int)blockIdx.x - wavesA) * 256 + threadIdx.x;
  if (idx < nU) {
    const int r = idx / H, h = idx - r * H;
    float acc = 0.f;
    for (int c = 0; c < Cin; ++c) acc += T1[((int64_t)c * R + r) * H + h] * A[c * R + r];
    gU[idx] = acc;
  } else if (idx < nU + nV) {
    const int j = idx - nU, r = j / W, w = j - r * W;
    float acc = 0.f;
    for (int c = 0; c < Cin; ++c) acc += T2[((int64_t)c * R + r) * W + w] * A[c * R + r];
    gV[j] = acc;
  }
}

// ------------------------------------------------------------------ activation
// O16 (round 6, bf16-mixed mode, BWD only): out is a bf16 tensor - d(pre-activation) for the layer's two gradient GEMMs, which
// round that operand to bf16 (to nearest even, as here) when they load it from fp32 words: same values, half the bytes thrice.
template <bool BWD, bool O16 = false>
__global__ void __launch_bounds__(256)
act_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ out,
           int64_t n, int act, bool vec) {
  if constexpr (O16) {
    const int64_t n4 = n >> 2;      // (the host checks n % 4 == 0 and the alignment)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(x)[i], g = reinterpret_cast<const float4*>(gy)[i];
      const float o[4] = {g.x * act_grad(v.x, act), g.y * act_grad(v.y, act), g.z * act_grad(v.z, act), g.w * act_grad(v.w, act)};
      uint2 r;
      r.x = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)o[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)o[1]) << 16);
      r.y = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)o[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)o[3]) << 16);
      reinterpret_cast<uint2*>(out)[i] = r;
    }
    return;
  }
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 o;
      if (BWD) {
        const float4 g = reinterpret_cast<const float4*>(gy)[i];
        o = make_float4(g.x * act_grad(v.x, act), g.y * act_grad(v.y, act), g.z * act_grad(v.z, act),
                        g.w * act_grad(v.w, act));
      } else {
        o = make_float4(act_apply(v.x, act), act_apply(v.y, act), act_apply(v.z, act), act_apply(v.w, act));
      }
      reinterpret_cast<float4*>(out)[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const float o = BWD ? gy[i] * act_grad(x[i], act) : act_apply(x[i], act);
      out[i] = o;
    }
  }
}

__global__ void __launch_bounds__(256)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n,
           bool vec) {
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
      reinterpret_cast<float4*>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
      y[i] = a[i] + b[i];
  }
}

// y[b, i] = x[b, i] + m[i]   (standalone GlobalBias.forward, reference model/blocks.py:196)
__global__ void __launch_bounds__(256)
add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y,
                 int64_t per, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    y[i] = x[i] + m[i % per];
}

// out[c][r] = in[r][c]  (32x32 tiles through LDS; used once per forward per weight matrix)
__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// ------------------------------------------------------------------ gated blend
__global__ void __launch_bounds__(256)
gated_blend_fwd_kernel(const float* __restrict__ h, const float* __restrict__ adv,
                       const float* __restrict__ alpha, float* __restrict__ out, int C, int P) {
  // one workgroup per (b,c) plane
  const int c = blockIdx.x % C;
  const float g = 1.0f / (1.0f + expf(-alpha[c]));
  const int64_t base = (int64_t)blockIdx.x * P;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float hv = h[base + p];
    out[base + p] = fmaf(g, adv[base + p] - hv, hv);     // (the GEMM epilogue's gated form computes the same bits)
  }
}

__global__ void __launch_bounds__(256)
gated_blend_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ h,
                       const float* __restrict__ adv, const float* __restrict__ alpha,
                       float* __restrict__ gh, float* __restrict__ gadv, float* __restrict__ partial,
                       int C, int P) {
  __shared__ float red[4];
  const int c = blockIdx.x % C;
  const float g = 1.0f / (1.0f + expf(-alpha[c]));
  const int64_t base = (int64_t)blockIdx.x * P;
  float acc = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float go = gout[base + p];
    gh[base + p] = (1.0f - g) * go;
    gadv[base + p] = g * go;
    acc += go * (adv[base + p] - h[base + p]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// from_out: the partial sums were taken against the blended OUTPUT, sum gout (out - h) = sigmoid sum gout (adv - h)
__global__ void __launch_bounds__(256)
gated_blend_finish(const float* __restrict__ partial, const float* __restrict__ alpha,
                   float* __restrict__ galpha, int B, int C, int from_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += partial[(int64_t)b * C + c];
  const float g = 1.0f / (1.0f + expf(-alpha[c]));
  galpha[c] = from_out ? s * (1.0f - g) : s * g * (1.0f - g);
}

// ------------------------------------------------------------------ bias / bias-map gradients
// grid (C, pchunks): gmap[c,p] = sum_b dz[b,c,p];  gbias[c] += sum over the chunk
__global__ void __launch_bounds__(256)
bias_grads_kernel(const float* __restrict__ dz, float* __restrict__ gmap, float* __restrict__ gbias,
                  int B, int C, int P, int64_t bs, int pchunks) {
  __shared__ float red[4];
  const int c = blockIdx.x / pchunks, chunk = blockIdx.x - c * pchunks;
  float acc = 0.f;
  for (int p = chunk * 256 + threadIdx.x; p < P; p += pchunks * 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dz[(int64_t)b * bs + (int64_t)c * P + p];
    if (gmap) gmap[(int64_t)c * P + p] = s;
    acc += s;
  }
  if (!gbias) return;
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&gbias[c], red[0] + red[1] + red[2] + red[3]);
}

// the same with 16-byte accesses (P % 4 == 0, 16-byte aligned rows): grid (C, pchunks) over float4 columns
__global__ void __launch_bounds__(256)
bias_grads_vec4_kernel(const float* __restrict__ dz, float* __restrict__ gmap, float* __restrict__ gbias,
                       int B, int C, int P4, int64_t bs, int pchunks) {
  __shared__ float red[4];
  const int c = blockIdx.x / pchunks, chunk = blockIdx.x - c * pchunks;
  const float4* src = reinterpret_cast<const float4*>(dz + (int64_t)c * P4 * 4);
  const int64_t bs4 = bs / 4;
  float acc = 0.f;
  for (int q = chunk * 256 + threadIdx.x; q < P4; q += pchunks * 256) {
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= B; b += 8) {     // eight independent loads in flight, summed in batch order
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(b + j) * bs4 + q];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; b < B; ++b) {
      const float4 v = src[(int64_t)b * bs4 + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (gmap) reinterpret_cast<float4*>(gmap + (int64_t)c * P4 * 4)[q] = s;
    acc += (s.x + s.y) + (s.z + s.w);
  }
  if (!gbias) return;
  acc = wave_sum_dpp(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&gbias[c], red[0] + red[1] + red[2] + red[3]);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void launch_gm8(const float* Pw, const float* gmap, float* gm8, int Cin, int Co, int64_t P, hipStream_t st) {   // Cin <= 16
  const dim3 grid((unsigned)((P + 63) / 64)), block(64 * GM8_SLICES);
  if (Cin <= 8) hipLaunchKernelGGL(gbias_gm8_kernel<8>, grid, block, 0, st, Pw, gmap, gm8, Cin, Co, P);
  else hipLaunchKernelGGL(gbias_gm8_kernel<16>, grid, block, 0, st, Pw, gmap, gm8, Cin, Co, P);
}
#ifndef GBIAS_GPW_ROWS           // (A/B builds: 0 = one workgroup per (o, c) on every grid)
#define GBIAS_GPW_ROWS 1
#endif
#ifndef GBIAS_GPW_ROWS_MINP      // smallest pixel count that takes the rows kernel
#define GBIAS_GPW_ROWS_MINP 2048
#endif
void launch_gpw(const float* gmap, const float* m8, float* gPw, int Cin, int Co, int64_t P, hipStream_t st) {
  if (GBIAS_GPW_ROWS && Cin <= 8 && P % 4 == 0 && P >= GBIAS_GPW_ROWS_MINP && aligned16(gmap) && aligned16(m8)) {
    if (Co >= 1024)
      hipLaunchKernelGGL((gbias_gpw_rows_kernel<4, 8>), dim3((Co + 3) / 4), dim3(256), 0, st, gmap, m8, gPw, Cin, Co, P / 4);
    else
      hipLaunchKernelGGL((gbias_gpw_rows_kernel<2, 8>), dim3((Co + 1) / 2), dim3(256), 0, st, gmap, m8, gPw, Cin, Co, P / 4);
    return;
  }
  hipLaunchKernelGGL(gbias_gpw_kernel, dim3(Co * Cin), dim3(256), 0, st, gmap, m8, gPw, Cin, P);
}

}  // namespace

extern "C" int paradis_global_bias_map_fwd(const float* A, const float* U, const float* V,
                                           const float* Pw, float* m8, float* map, int Cin, int Co,
                                           int R, int H, int W, void* stream) {
  PD_REQUIRE(Cin >= 1 && Co >= 1 && R >= 1 && H >= 1 && W >= 1, "global_bias_map_fwd: bad shape");
  PD_REQUIRE(Pw != nullptr || Cin == Co, "global_bias_map_fwd: projection required when Cin != Co");
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = (int64_t)H * W;
  float* m8_dst = Pw ? m8 : map;
  PD_REQUIRE(H <= 65535 && Cin <= 65535 && Co <= 65535, "global_bias_map_fwd: grid too large");
  if ((int64_t)H * W >= (1 << 17) && R <= GM8_MAXR)
    hipLaunchKernelGGL(gbias_m8_rows_kernel<8>, dim3((W + 63) / 64, (H + 7) / 8, Cin), dim3(256), 0, st, A, U, V, m8_dst,
                       Cin, R, H, W);
  else
    hipLaunchKernelGGL(gbias_m8_kernel, dim3((W + 63) / 64, H, Cin), dim3(256), 0, st, A, U, V, m8_dst, Cin, R, H, W);
  if (Pw)
    hipLaunchKernelGGL(small_mix_kernel, dim3((unsigned)((P + 255) / 256), Co), dim3(256), 0, st, Pw, Cin, 1,
                       m8, map, Co, Cin, P);
  PD_CHECK_LAUNCH("global_bias_map_fwd");
  return 0;
}

extern "C" size_t paradis_global_bias_map_bwd_ws_bytes(int Cin, int Co, int R, int H, int W) {
  (void)Co;
  return ((size_t)Cin * H * W + (size_t)Cin * R * H + (size_t)Cin * R * W) * sizeof(float) + 256;
}

extern "C" int paradis_global_bias_map_bwd(const float* gmap, const float* A, const float* U,
                                           const float* V, const float* Pw, const float* m8, float* gA,
                                           float* gU, float* gV, float* gPw, int Cin, int Co, int R,
                                           int H, int W, void* workspace, void* stream) {
  PD_REQUIRE(Cin >= 1 && Co >= 1 && R >= 1 && H >= 1 && W >= 1, "global_bias_map_bwd: bad shape");
  PD_REQUIRE(workspace != nullptr, "global_bias_map_bwd: workspace required");
  PD_REQUIRE(Pw != nullptr || Cin == Co, "global_bias_map_bwd: projection required when Cin != Co");
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = (int64_t)H * W;
  float* gm8 = (float*)workspace;
  float* T1 = gm8 + (size_t)Cin * P;
  float* T2 = T1 + (size_t)Cin * R * H;
  const float* gm8_src = gmap;
  if (Pw) {
    PD_REQUIRE(m8 != nullptr && gPw != nullptr, "global_bias_map_bwd: m8/gPw required with projection");
    launch_gpw(gmap, m8, gPw, Cin, Co, P, st);
    // gm8[c,p] = sum_o Pw[o,c] gmap[o,p]
    if (Cin <= 16) {
      launch_gm8(Pw, gmap, gm8, Cin, Co, P, st);
    } else {
      hipLaunchKernelGGL(small_mix_kernel, dim3((unsigned)((P + 255) / 256), Cin), dim3(256), 0, st, Pw, 1,
                         Cin, gmap, gm8, Cin, Co, P);
    }
    gm8_src = gm8;
  }
  // (T1, T2 and the finish in ONE kernel - everything is local to a rank index r: R workgroups walking gm8 once, row dot
  //  products by wave sums, column sums in registers - was built in round 5: 31.5 us per call in the 32x64 step against
  //  9.1 + 10.5 + 9.1 us for the three wide kernels, 3.2 against 1.56 ms at 721x1440; removed)
  const int64_t nwaves = (int64_t)Cin * R * H;
  hipLaunchKernelGGL(gbias_t1_kernel, dim3((unsigned)((nwaves * 64 + 255) / 256)), dim3(256), 0, st,
                     gm8_src, V, T1, Cin, R, H, W);
  hipLaunchKernelGGL(gbias_t2_kernel, dim3(stream_blocks((int64_t)Cin * R * W)), dim3(256), 0, st, gm8_src,
                     U, T2, Cin, R, H, W);
  const int nfin = (Cin * R + 3) / 4 + (R * H + R * W + 255) / 256;      // gA waves, then gU / gV threads
  hipLaunchKernelGGL(gbias_finish_kernel, dim3(nfin), dim3(256), 0, st, T1, T2, A, U, gA, gU,
                     gV, Cin, R, H, W);
  PD_CHECK_LAUNCH("global_bias_map_bwd");
  return 0;
}

extern "C" int paradis_act_fwd(const float* x, float* y, int64_t n, int act, void* stream) {
  PD_REQUIRE(n >= 0 && act >= 0 && act <= 2, "act_fwd: bad arguments");
  if (n == 0) return 0;
  const bool vec = (n % 4 == 0) && aligned16(x) && aligned16(y);
  hipLaunchKernelGGL(act_kernel<false>, dim3(stream_blocks(vec ? n / 4 : n)), dim3(256), 0,
                     (hipStream_t)stream, nullptr, x, y, n, act, vec);
  PD_CHECK_LAUNCH("act_fwd");
  return 0;
}

extern "C" int paradis_act_bwd(const float* gy, const float* x, float* gx, int64_t n, int act, void* stream) {
  PD_REQUIRE(n >= 0 && act >= 0 && act <= 2, "act_bwd: bad arguments");
  if (n == 0) return 0;
  const bool vec = (n % 4 == 0) && aligned16(x) && aligned16(gy) && aligned16(gx);
  hipLaunchKernelGGL(act_kernel<true>, dim3(stream_blocks(vec ? n / 4 : n)), dim3(256), 0,
                     (hipStream_t)stream, gy, x, gx, n, act, vec);
  PD_CHECK_LAUNCH("act_bwd");
  return 0;
}

// gx as a bf16 tensor (ABI 9; bf16-mixed mode: gx = d(pre-activation), the operand of the layer's gradient GEMMs).  n % 4 == 0,
// 16-byte aligned gy / x, 8-byte aligned gx.
extern "C" int paradis_act_bwd16(const float* gy, const float* x, void* gx, int64_t n, int act, void* stream) {
  PD_REQUIRE(n >= 0 && act >= 0 && act <= 2, "act_bwd16: bad arguments");
  if (n == 0) return 0;
  PD_REQUIRE(n % 4 == 0 && aligned16(x) && aligned16(gy) && (reinterpret_cast<uintptr_t>(gx) & 7) == 0,
             "act_bwd16: n %% 4 == 0 and aligned tensors required");
  hipLaunchKernelGGL((act_kernel<true, true>), dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, gy, x,
                     (float*)gx, n, act, true);
  PD_CHECK_LAUNCH("act_bwd16");
  return 0;
}

extern "C" int paradis_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  PD_REQUIRE(n >= 0, "add: bad arguments");
  if (n == 0) return 0;
  const bool vec = (n % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(y);
  hipLaunchKernelGGL(add_kernel, dim3(stream_blocks(vec ? n / 4 : n)), dim3(256), 0, (hipStream_t)stream,
                     a, b, y, n, vec);
  PD_CHECK_LAUNCH("add");
  return 0;
}

extern "C" int paradis_gated_blend_fwd(const float* h, const float* adv, const float* alpha,
                                       float* out, int B, int C, int P, void* stream) {
  PD_REQUIRE(B >= 0 && C >= 1 && P >= 1, "gated_blend_fwd: bad shape");
  if (B == 0) return 0;
  hipLaunchKernelGGL(gated_blend_fwd_kernel, dim3((unsigned)((int64_t)B * C)), dim3(256), 0,
                     (hipStream_t)stream, h, adv, alpha, out, C, P);
  PD_CHECK_LAUNCH("gated_blend_fwd");
  return 0;
}

extern "C" size_t paradis_gated_blend_bwd_ws_bytes(int B, int C, int P) {
  (void)P;
  return (size_t)std::max(B, 1) * C * sizeof(float) + 256;
}

static int gated_blend_bwd_impl(const float* gout, const float* h, const float* third, const float* alpha, float* gh,
                                float* gadv, float* galpha, int B, int C, int P, void* workspace, void* stream,
                                int from_out) {
  PD_REQUIRE(B >= 0 && C >= 1 && P >= 1, "gated_blend_bwd: bad shape");
  PD_REQUIRE(workspace != nullptr, "gated_blend_bwd: workspace required");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (B > 0)
    hipLaunchKernelGGL(gated_blend_bwd_kernel, dim3((unsigned)((int64_t)B * C)), dim3(256), 0, st, gout, h,
                       third, alpha, gh, gadv, partial, C, P);
  hipLaunchKernelGGL(gated_blend_finish, dim3((C + 255) / 256), dim3(256), 0, st, partial, alpha, galpha, B, C,
                     from_out);
  PD_CHECK_LAUNCH("gated_blend_bwd");
  return 0;
}

extern "C" int paradis_gated_blend_bwd(const float* gout, const float* h, const float* adv,
                                       const float* alpha, float* gh, float* gadv, float* galpha, int B,
                                       int C, int P, void* workspace, void* stream) {
  return gated_blend_bwd_impl(gout, h, adv, alpha, gh, gadv, galpha, B, C, P, workspace, stream, 0);
}

// bias / bias-map gradients on a bf16-STORED dz (bf16-mixed mode, round 6): eight columns per thread from one 16-byte load per sample
// (P % 8 == 0, 16-byte aligned rows); fp32 sums in batch order, fp32 outputs
__global__ void __launch_bounds__(256)
bias_grads_b16_kernel(const uint16_t* __restrict__ dz, float* __restrict__ gmap, float* __restrict__ gbias,
                      int B, int C, int P8, int64_t bs, int pchunks) {
  __shared__ float red[4];
  const int c = blockIdx.x / pchunks, chunk = blockIdx.x - c * pchunks;
  const uint4* src = reinterpret_cast<const uint4*>(dz + (int64_t)c * P8 * 8);
  const int64_t bs8 = bs / 8;
  float acc = 0.f;
  for (int q = chunk * 256 + threadIdx.x; q < P8; q += pchunks * 256) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 8) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)min(b0 + j, B - 1) * bs8 + q];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (b0 + j < B) {
          const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s[2 * i] += __uint_as_float(w[i] << 16);
            s[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u);
          }
        }
      }
    }
    if (gmap) {
      float4* o = reinterpret_cast<float4*>(gmap + (int64_t)c * P8 * 8) + 2 * q;
      o[0] = make_float4(s[0], s[1], s[2], s[3]);
      o[1] = make_float4(s[4], s[5], s[6], s[7]);
    }
    acc += ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  if (!gbias) return;
  acc = wave_sum_dpp(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&gbias[c], red[0] + red[1] + red[2] + red[3]);
}

// The same gradients when the advected tensor was never materialised (paradis_pw_gemm_fwd_gated): `out` is the
// blended output, adv - h = (out - h) / sigmoid, so galpha = (1 - sigmoid) sum gout (out - h) - no division.
extern "C" int paradis_gated_blend_bwd_out(const float* gout, const float* h, const float* out,
                                           const float* alpha, float* gh, float* gadv, float* galpha, int B,
                                           int C, int P, void* workspace, void* stream) {
  return gated_blend_bwd_impl(gout, h, out, alpha, gh, gadv, galpha, B, C, P, workspace, stream, 1);
}

extern "C" int paradis_bias_grads(const float* dz, float* gmap, float* gbias, int B, int C, int P,
                                  int64_t dz_bs, void* stream) {
  PD_REQUIRE(B >= 0 && C >= 1 && P >= 1, "bias_grads: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (gbias && pd_zero_async(gbias, (size_t)C * sizeof(float), st) != hipSuccess) {
    paradis_set_error("bias_grads: memset failed");
    return 2;
  }
  if (!gmap && !gbias) return 0;
  if (P % 4 == 0 && dz_bs % 4 == 0 && aligned16(dz) && (!gmap || aligned16(gmap))) {
    const int P4 = P / 4;
    // at most TWO chunks per channel: their atomic adds into the zeroed gbias commute exactly (a + b = b + a), so the
    // result does not depend on which workgroup arrives first
    int pch = std::max(1, std::min(2, std::min((P4 + 255) / 256, std::max(1, 2048 / C))));
    if (paradis_deterministic()) pch = 1;
    hipLaunchKernelGGL(bias_grads_vec4_kernel, dim3((unsigned)((int64_t)C * pch)), dim3(256), 0, st, dz, gmap,
                       gbias, B, C, P4, dz_bs, pch);
    PD_CHECK_LAUNCH("bias_grads");
    return 0;
  }
  int pchunks = std::max(1, std::min(2, std::min((P + 255) / 256, std::max(1, 2048 / C))));   // (two adds commute)
  if (paradis_deterministic()) pchunks = 1;   // one workgroup per channel: no atomics between chunks
  hipLaunchKernelGGL(bias_grads_kernel, dim3((unsigned)((int64_t)C * pchunks)), dim3(256), 0, st, dz, gmap,
                     gbias, B, C, P, dz_bs, pchunks);
  PD_CHECK_LAUNCH("bias_grads");
  return 0;
}

// dz stored as bf16 [B][C,P] (bf16-mixed mode): same outputs, fp32
extern "C" int paradis_bias_grads16(const void* dz, float* gmap, float* gbias, int B, int C, int P,
                                    int64_t dz_bs, void* stream) {
  PD_REQUIRE(B >= 0 && C >= 1 && P >= 1, "bias_grads16: bad shape");
  PD_REQUIRE(P % 8 == 0 && dz_bs % 8 == 0 && aligned16(dz) && (!gmap || aligned16(gmap)),
             "bias_grads16: needs P %% 8 == 0 and 16-byte aligned rows");
  hipStream_t st = (hipStream_t)stream;
  if (gbias && pd_zero_async(gbias, (size_t)C * sizeof(float), st) != hipSuccess) {
    paradis_set_error("bias_grads16: memset failed");
    return 2;
  }
  if ((!gmap && !gbias) || B == 0) {
    if (gmap && pd_zero_async(gmap, (size_t)C * P * sizeof(float), st) != hipSuccess) return 2;
    return 0;
  }
  const int P8 = P / 8;
  int pch = std::max(1, std::min(2, std::min((P8 + 255) / 256, std::max(1, 2048 / C))));   // (two adds commute)
  if (paradis_deterministic()) pch = 1;
  hipLaunchKernelGGL(bias_grads_b16_kernel, dim3((unsigned)((int64_t)C * pch)), dim3(256), 0, st,
                     reinterpret_cast<const uint16_t*>(dz), gmap, gbias, B, C, P8, dz_bs, pch);
  PD_CHECK_LAUNCH("bias_grads16");
  return 0;
}

extern "C" int paradis_add_bcast(const float* x, const float* m, float* y, int64_t per_sample, int B,
                                 void* stream) {
  PD_REQUIRE(per_sample >= 1 && B >= 0, "add_bcast: bad arguments");
  if (B == 0) return 0;
  const int64_t total = per_sample * B;
  hipLaunchKernelGGL(add_bcast_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, m,
                     y, per_sample, total);
  PD_CHECK_LAUNCH("add_bcast");
  return 0;
}

extern "C" int paradis_transpose(const float* in, float* out, int rows, int cols, void* stream) {
  PD_REQUIRE(rows >= 1 && cols >= 1, "transpose: bad shape");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0,
                     (hipStream_t)stream, in, out, rows, cols);
  PD_CHECK_LAUNCH("transpose");
  return 0;
}

// ---- GlobalBias in two stages (lets the GEMM epilogue apply the projection on the fly) ----
// stage 1: m8[Cin,H,W] = sum_r A[c,r] U[r,h] V[r,w]
extern "C" int paradis_global_bias_m8_fwd(const float* A, const float* U, const float* V, float* m8,
                                          int Cin, int R, int H, int W, void* stream) {
  return paradis_global_bias_map_fwd(A, U, V, nullptr, nullptr, m8, Cin, Cin, R, H, W, stream);
}
// adjoint of stage 1: (gA, gU, gV) from gm8; workspace >= paradis_global_bias_map_bwd_ws_bytes
extern "C" int paradis_global_bias_m8_bwd(const float* gm8, const float* A, const float* U,
                                          const float* V, float* gA, float* gU, float* gV, int Cin,
                                          int R, int H, int W, void* workspace, void* stream) {
  return paradis_global_bias_map_bwd(gm8, A, U, V, nullptr, nullptr, gA, gU, gV, nullptr, Cin, Cin, R, H,
                                     W, workspace, stream);
}
// adjoint of the projection map[o,p] = sum_c Pw[o,c] m8[c,p]:  gPw[o,c] = sum_p gmap[o,p] m8[c,p],
// gm8[c,p] = sum_o Pw[o,c] gmap[o,p]
extern "C" int paradis_global_bias_proj_bwd(const float* gmap, const float* m8, const float* Pw,
                                            float* gPw, float* gm8, int Cin, int Co, int64_t P,
                                            void* stream) {
  PD_REQUIRE(Cin >= 1 && Cin <= 16 && Co >= 1 && P >= 1, "global_bias_proj_bwd: bad shape (Cin <= 16)");
  hipStream_t st = (hipStream_t)stream;
  launch_gpw(gmap, m8, gPw, Cin, Co, P, st);
  launch_gm8(Pw, gmap, gm8, Cin, Co, P, st);
  PD_CHECK_LAUNCH("global_bias_proj_bwd");
  return 0;
}
