// f3 (second half): Muon / NorMuon step for one 2-D weight matrix, without Triton.
// The reference's default optimiser (config/paradis_settings.yaml:117, trainer.py:337-364) comes from
// the un-vendored, un-pinned package `dion` (requirements.txt:26, git HEAD of microsoft/dion); this
// file restates its PUBLISHED algorithm (PARITY UNPINNED - no reference outputs can be generated here):
//   M <- mu M + G;  U = M (or G + mu M with Nesterov)
//   X = U / (||U||_F + eps), transposed so that rows <= cols
//   5 x quintic Newton-Schulz:  A = X X^T;  X <- a X + (b A + c A A) X      (dion's per-step a,b,c)
//   NorMuon only: v <- beta2 v + (1-beta2) rowmean(X^2);  X <- X / (sqrt(v)+1e-8), rescaled to the
//                 Frobenius norm it had before
//   W <- W (1 - lr wd) - lr_adj X
// dion runs the iteration in bf16 through Triton; here it is fp32 on the GEMMs of gemm.hip (exact f32 MFMA
// or the bf16-split arithmetic, as the caller's pointwise GEMMs)
// (a X + B X is computed as (B + a I) X, so the GEMM needs no scaled epilogue).  All scalars (norms)
// stay on the device: no host synchronisation.
#include <math.h>
#include "common.h"

extern "C" int paradis_bgemm(const float* A, const float* AT, const float* Bm, float* C, int nbatch, int M,
                             int K, int N, int64_t a_bs, int64_t at_bs, int64_t b_bs, int64_t c_bs,
                             void* split_ws, void* stream);


namespace {

constexpr float NS_A[5] = {4.0848f, 3.9505f, 3.7418f, 2.8769f, 2.8366f};
constexpr float NS_B[5] = {-6.8946f, -6.3029f, -5.5913f, -3.1427f, -3.0525f};
constexpr float NS_C[5] = {2.9270f, 2.6377f, 2.3037f, 1.2046f, 1.2012f};

// Same-shaped matrices are processed together (blockIdx.y = matrix t): a single 896^3 product is 49
// output tiles on a 256-CU chip, a stack of 16 fills it.  tab = device table [4][stride] of the
// addresses of w, g, m, v of the T matrices of the group.
struct Tab {
  const int64_t* p; int stride;
  __device__ __forceinline__ float* w(int t) const { return reinterpret_cast<float*>(p[t]); }
  __device__ __forceinline__ const float* g(int t) const { return reinterpret_cast<const float*>(p[stride + t]); }
  __device__ __forceinline__ float* m(int t) const { return reinterpret_cast<float*>(p[2 * stride + t]); }
  __device__ __forceinline__ float* v(int t) const { return reinterpret_cast<float*>(p[3 * stride + t]); }
};

// m = mu m + g ; u = nesterov ? g + mu m : m ; sumsq[t] += sum u^2   (sumsq pre-zeroed)
__global__ void __launch_bounds__(256)
muon_momentum_kernel(Tab tab, float* __restrict__ U, int64_t n, float mu, int nesterov,
                     float* __restrict__ sumsq) {
  __shared__ float red[4];
  const int t = blockIdx.y;
  float* m = tab.m(t);
  const float* g = tab.g(t);
  float* u = U + (int64_t)t * n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = mu * m[i] + gi;
    m[i] = mi;
    const float ui = nesterov ? gi + mu * mi : mi;
    u[i] = ui;
    acc += ui * ui;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sumsq[t], red[0] + red[1] + red[2] + red[3]);
}

// y_t = x_t / (sqrt(sumsq[t]) + eps)
__global__ void __launch_bounds__(256)
muon_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                      const float* __restrict__ sumsq, float eps) {
  const int t = blockIdx.y;
  const float inv = 1.0f / (sqrtf(sumsq[t]) + eps);
  x += (int64_t)t * n; y += (int64_t)t * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * inv;
}

// out_t[c][r] = in_t[r][c]   (rows x cols -> cols x rows), 32x32 tiles through LDS
__global__ void __launch_bounds__(256)
muon_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int64_t n = (int64_t)rows * cols;
  in += (int64_t)blockIdx.z * n; out += (int64_t)blockIdx.z * n;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(int64_t)(r0 + j) * cols + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < cols && r0 + tx < rows) out[(int64_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// out_t = b A_t + c A2_t + a I   (M x M)
__global__ void __launch_bounds__(256)
muon_poly_kernel(const float* __restrict__ A, const float* __restrict__ A2, float* __restrict__ out, int M,
                 float a, float b, float c) {
  const int64_t n = (int64_t)M * M, o = (int64_t)blockIdx.y * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / M), col = (int)(i - (int64_t)r * M);
    out[o + i] = b * A[o + i] + c * A2[o + i] + (r == col ? a : 0.f);
  }
}

// NorMuon, one block per neuron (row of the un-transposed matrix) of matrix t = blockIdx.y:
// v[r] = beta2 v[r] + (1-beta2) mean_c x[r,c]^2 ; x[r,:] /= sqrt(v[r]) + 1e-8 ;
// sums[2t] += sum x_old^2 ; sums[2t+1] += sum x_new^2
__global__ void __launch_bounds__(256)
normuon_rows_kernel(float* __restrict__ X, Tab tab, int rows, int cols, int64_t rs, int64_t cs, float beta2,
                    float* __restrict__ sums) {
  __shared__ float red[4];
  __shared__ float stat;
  const int r = blockIdx.x, t = blockIdx.y;
  float* x = X + (int64_t)t * rows * cols;
  float* v = tab.v(t);
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float e = x[(int64_t)r * rs + (int64_t)c * cs];
    acc += e * e;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ss = red[0] + red[1] + red[2] + red[3];
    const float vn = beta2 * v[r] + (1.0f - beta2) * (ss / (float)cols);
    v[r] = vn;
    const float inv = 1.0f / (sqrtf(vn) + 1e-8f);
    stat = inv;
    atomicAdd(&sums[2 * t], ss);
    atomicAdd(&sums[2 * t + 1], ss * inv * inv);
  }
  __syncthreads();
  const float inv = stat;
  for (int c = threadIdx.x; c < cols; c += 256) x[(int64_t)r * rs + (int64_t)c * cs] *= inv;
}

// w_t = w_t (1 - lr wd) - alpha ratio_t u_t,  ratio = sqrt(sums[2t]) / max(sqrt(sums[2t+1]), 1e-8) when
// sums given; u is read with strides (it may be held transposed)
__global__ void __launch_bounds__(256)
muon_apply_kernel(Tab tab, const float* __restrict__ U, int rows, int cols, int64_t urs, int64_t ucs,
                  float decay, float alpha, const float* __restrict__ sums) {
  const int t = blockIdx.y;
  float ratio = 1.0f;
  if (sums) ratio = sqrtf(sums[2 * t]) / fmaxf(sqrtf(sums[2 * t + 1]), 1e-8f);
  const float k = alpha * ratio;
  const int64_t n = (int64_t)rows * cols;
  float* w = tab.w(t);
  const float* u = U + (int64_t)t * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    w[i] = w[i] * decay - k * u[(int64_t)r * urs + (int64_t)c * ucs];
  }
}

inline int blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024)); }
inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

}  // namespace

// bytes of workspace for a group of T matrices [rows, cols]
extern "C" size_t paradis_muon_ws_bytes(int T, int rows, int cols) {
  if (T <= 0 || rows <= 0 || cols <= 0) return 256;
  const size_t n = (size_t)rows * cols, m = (size_t)std::min(rows, cols);
  // U, X, XT, Xnew (T x n each), A, A2, B' (T x m x m each), scalars; then the bf16-split images of the
  // left operands of the Newton-Schulz products (T x [m, max(rows, cols)], 256-B aligned)
  const size_t floats = (up64(n * T)) * 4 + up64(m * m * T) * 3 + up64(3 * (size_t)T) + 64;
  return floats * sizeof(float) + (size_t)T * paradis_pw_gemm_split_bytes((int)m, std::max(rows, cols), PARADIS_GEMM_BF16X3) + 256;
}

// One Muon (normuon = 0) or NorMuon (normuon = 1) step on T same-shaped weight matrices w_t[rows, cols]
// (conv weights flattened to [out, in*kh*kw] by the caller).  ptrs: DEVICE table [4][table_stride] of
// the addresses of w, g, m (momentum, [rows*cols]) and v (NorMuon per-row state [rows]; unused for
// Muon), this group's entries first.  lr_adj = the shape-adjusted learning rate.
extern "C" int paradis_muon_step(const int64_t* ptrs, int table_stride, int T, int rows, int cols, float lr,
                                 float lr_adj, float mu, float beta2, float weight_decay, float eps,
                                 int nesterov, int normuon, int split, void* workspace, void* stream) {
  PD_REQUIRE(T >= 0 && rows >= 1 && cols >= 1 && table_stride >= T, "muon_step: bad shape");
  if (T == 0) return 0;
  PD_REQUIRE(ptrs != nullptr && workspace != nullptr, "muon_step: table / workspace required");
  PD_REQUIRE(T <= 65535, "muon_step: too many matrices in one group");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)rows * cols;
  const bool tr = rows > cols;              // iterate on the wide orientation
  const int M = tr ? cols : rows, K = tr ? rows : cols;
  const int64_t mm = (int64_t)M * M;
  Tab tab{ptrs, table_stride};
  float* ws = (float*)workspace;
  float* U = ws;                 ws += up64((size_t)n * T);
  float* X = ws;                 ws += up64((size_t)n * T);
  float* XT = ws;                ws += up64((size_t)n * T);
  float* Xn = ws;                ws += up64((size_t)n * T);
  float* A = ws;                 ws += up64((size_t)mm * T);
  float* A2 = ws;                ws += up64((size_t)mm * T);
  float* Bp = ws;                ws += up64((size_t)mm * T);
  float* sc = ws;                // [T] sum u^2, then [2T] NorMuon sums
  ws += up64(3 * (size_t)T) + 64;
  // split != 0: the three products of every iteration run on the bf16-split GEMM (gemm.hip); the images
  // of their left operands ([M,K] or [M,M], K >= M) live behind the float workspace
  void* img = split ? (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
  if (pd_zero_async(sc, 3 * (size_t)T * sizeof(float), st) != hipSuccess) {
    paradis_set_error("muon_step: memset failed");
    return 2;
  }
  const dim3 gn(blocks(n), T);
  hipLaunchKernelGGL(muon_momentum_kernel, gn, dim3(256), 0, st, tab, U, n, mu, nesterov, sc);
  // X = U / (||U|| + eps), in the wide orientation [M, K]
  auto transpose = [&](const float* in, float* out, int r, int c) {
    hipLaunchKernelGGL(muon_transpose_kernel, dim3((c + 31) / 32, (r + 31) / 32, T), dim3(256), 0, st, in, out, r, c);
  };
  if (tr) {
    hipLaunchKernelGGL(muon_normalize_kernel, gn, dim3(256), 0, st, (const float*)U, XT, n, (const float*)sc, eps);
    transpose(XT, X, rows, cols);                                                                       // X [M, K]
  } else {
    hipLaunchKernelGGL(muon_normalize_kernel, gn, dim3(256), 0, st, (const float*)U, X, n, (const float*)sc, eps);
  }
  float* cur = X;
  float* nxt = Xn;
  for (int it = 0; it < 5; ++it) {
    transpose(cur, XT, M, K);                                                                           // XT [K, M]
    // A = X X^T : [M,K] x [K,M].  The transposed left operand each product needs for the LDS-DMA kernel
    // is at hand: X^T here, and A and B' below are symmetric.
    if (int e = paradis_bgemm(cur, XT, XT, A, T, M, K, M, n, n, n, mm, img, stream)) return e;
    if (int e = paradis_bgemm(A, A, A, A2, T, M, M, M, mm, mm, mm, mm, img, stream)) return e;
    hipLaunchKernelGGL(muon_poly_kernel, dim3(blocks(mm), T), dim3(256), 0, st, (const float*)A, (const float*)A2,
                       Bp, M, NS_A[it], NS_B[it], NS_C[it]);
    // X <- (b A + c A^2 + a I) X : [M,M] x [M,K]
    if (int e = paradis_bgemm(Bp, Bp, cur, nxt, T, M, M, K, mm, mm, n, n, img, stream)) return e;
    std::swap(cur, nxt);
  }
  // cur = orthogonalised updates in the wide orientation; element (r, c) of the [rows, cols] matrix is
  // cur[c*K + r] when transposed, cur[r*K + c] otherwise
  const int64_t urs = tr ? 1 : K, ucs = tr ? K : 1;
  if (normuon)
    hipLaunchKernelGGL(normuon_rows_kernel, dim3(rows, T), dim3(256), 0, st, cur, tab, rows, cols, urs, ucs, beta2,
                       sc + T);
  hipLaunchKernelGGL(muon_apply_kernel, gn, dim3(256), 0, st, tab, (const float*)cur, rows, cols, urs, ucs,
                     1.0f - lr * weight_decay, lr_adj, normuon ? (const float*)(sc + T) : (const float*)nullptr);
  PD_CHECK_LAUNCH("muon_step");
  return 0;
}
