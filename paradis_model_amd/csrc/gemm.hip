// a6: pointwise (1x1) channel mixing as a batched GEMM on FP32 MFMA (v_mfma_f32_32x32x2_f32).
// Reference call sites: model/blocks.py:86 (CLinear), :110 (SepConv pointwise).
//
//   fwd   : Y[b][Co,P] = epi( W[Co,Ci] . X[b][Ci,P] )      A = W  (k-contiguous), B = X  (n-contiguous)
//   dgrad : dX[b][Ci,P] = epi( W^T . dY[b][Co,P] )         A = W^T(m-contiguous), B = dY (n-contiguous)
//   wgrad : dW[Co,Ci]   = sum_b dY[b] . X[b]^T             A = dY (k-contiguous), B = X^T(k-contiguous)
//
// Exact fp32: the f32 MFMA is a k-ordered fmaf chain (no TF32/xf32 on gfx950), so results differ
// from the CPU's blocked SGEMM only by summation order.
//
// Tile: 128x128x32 per 256-thread workgroup; wave (wm,wn) owns 64x64 = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs).  Both operands are staged through registers into LDS as [k][m|n] images
// (row pitch 129 for transposing stores, 132 for vector stores: both conflict-free) so that
// fragment reads are conflict-free ds_read_b32; two LDS stages (66 KiB => exactly 2 workgroups per
// CU, which makes every shape of this model an integral number of rounds over the 256 CUs), one
// barrier per k-tile, global loads for tile t+1 in flight during the 64 MFMAs of tile t, fragment
// reads for k-step kk+1 issued before the MFMAs of kk.  Work-group ids are remapped so that the
// M-tiles that share one X tile run on the same XCD (L2 reuse of X).
#include <algorithm>
#include <cstdlib>
#include "common.h"

// Three translation units from this one source (the Makefile compiles it three times; -DGEMM_PART unset = all in one):
//   GEMM_PART == 1: the fp32-width schemes (exact f32 MFMA, bf16x3, f16x2) and every host entry point of the C ABI;
//   GEMM_PART == 2: the forward / data-gradient kernels of the bf16-mixed scheme (PARADIS_GEMM_BF16), pd_amp_launch_fwd;
//   GEMM_PART == 3: its weight-gradient kernels, pd_amp_launch_wgrad.
// All kernels are templates or sit behind the guards, so each unit instantiates only what it launches: a clean build
// takes the time of the slowest third on three cores instead of five minutes on one (the bf16-mixed kernels come in
// four I/O-type instantiations each, and each inlines the whole epilogue).
#ifndef GEMM_PART
#define GEMM_PART 0
#endif

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t a_bs, b_bs, c_bs;   // stride between grid batches (fwd/dgrad: sample; wgrad: split slab)
  int nbatch;                 // grid batches (fwd/dgrad: samples; wgrad: k-range splits)
  int inner;                  // wgrad: number of samples reduced (0 for fwd/dgrad)
  int64_t a_is, b_is;         // wgrad: strides between samples
  // epilogue:  v = acc (+bias[m]) (+map[m,n]); zout = v; v = zmul ? v*act'(zmul) : act(v);
  //            v = gate ? res + sigmoid(gate[m]) (v - res) : v + res
  const float* bias; const float* map; const float* res; const float* zmul; float* zout;
  const float* gate;          // [M] or NULL: the residual is blended in per output channel (gated blend of the advection)
  int64_t res_bs, zmul_bs, zout_bs;
  int act;
  int stagger;                // start-up skew between co-resident workgroups, in units of 512 cycles
  float* rowsum;              // wgrad only: [nbatch][M] partial row sums of A (= bias gradient), or NULL
  // low-rank bias map applied on the fly: acc[m,n] += sum_c pw[c*M + m] * m8[c*N + n]  (M % 4 == 0)
  const float* m8; const float* pw; int cin;
  // f16x2 scheme: where the operands' max |value| comes from.  a_amax: one word (bits of max |A|, weight
  // image tail) for fwd/dgrad, PARADIS_AMAX_PARTIALS words for wgrad; b_amax: PARADIS_AMAX_PARTIALS words.
  const uint32_t* a_amax; const uint32_t* b_amax;
  // PARADIS_GEMM_BF16 only (round 6): which tensors are STORED as bf16 (2 bytes per element; strides stay in elements).
  // IO_B16: the activation operand B (fwd: X, dgrad: dY) - pw_gemm_b16_kernel stages it by LDS-DMA and reads it with
  // ds_read_b64_tr_b16; IO_C16: the output C (and zout); IO_ZM16: zmul.  Residual, bias and maps are always fp32.
  int io16;
  // (PARADIS_GEMM_BF16, the reference's bf16-mixed mode: the result is rounded to bf16 where the reference's autocast
  //  conv2d rounds it - the pre-activation and the activated value (fwd), the activation-gradient product (dgrad) -
  //  before the fp32 residual / blend; a compile-time property of pw_gemm_bf16_k32_kernel's epilogue.  Stored as fp32.)
};

// launchers of the bf16-mixed kernels (defined in units 2 and 3)
int pd_amp_launch_fwd(const GemmArgs& d, hipStream_t st);
// kind: 0 = 128 x 128 tile, fp32 operands; 1 = 128 x 128 with a bf16 operand; 2 = 256 x 128; 3 = 256 x 256 (grid: the caller's)
int pd_amp_launch_wgrad(const GemmArgs& g, int io16, int kind, int grid, hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;
constexpr int ld_of(bool kc) { return kc ? BM + 1 : BM + 4; }  // floats per k-row of the LDS image
constexpr int stage_floats(int bk) { return bk * (BM + 4); }   // per operand per stage (upper bound)
constexpr size_t lds_bytes(int bk) { return (size_t)4 * stage_floats(bk) * sizeof(float); }
constexpr int nv_of(int bk) { return BM * bk / (256 * 4); }    // float4 loads per thread per operand

// Tunables (debug setters below; defaults chosen from tools/gemm_bench.py measurements)
#if GEMM_PART < 2
int g_bk = 16;            // k-tile depth: 16 or 32
int g_wg_per_cu = 4;      // resident workgroups per CU enforced through the dynamic-LDS request
int g_stagger = 0;        // see GemmArgs::stagger
int g_dma_stages = 3;     // LDS-DMA ring depth for row-contiguous operands (0 = never use the DMA kernel)
int g_wgrad_dma_stages = 2;  // LDS-DMA ring depth of the weight-gradient kernel (0 = register-staged)
#endif


// ---- staging: 128 x 16 operand slab -> registers -> LDS image [k][m] -------------------------
// KC: element (row=m|n, k) at base[row*ld + k]   (k contiguous)
// MC: element (row=m|n, k) at base[k*ld + row]   (row contiguous)
template <bool KC, int BK>
__device__ __forceinline__ void slab_load(const float* __restrict__ base, int64_t ld, int row0,
                                          int k0, int rows, int K, bool vec_ok,
                                          float4 (&r)[nv_of(BK)]) {
  constexpr int NV = nv_of(BK);
  constexpr int TPR = BK / 4;       // threads per row (k-contiguous layout)
  constexpr int RPP = 256 / TPR;    // rows per pass
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (KC) {
      const int row = row0 + (tid / TPR) + RPP * i, k = k0 + (tid % TPR) * 4;
      const float* p = base + (int64_t)row * ld + k;
      if (vec_ok && row < rows && k + 3 < K) {
        r[i] = *reinterpret_cast<const float4*>(p);
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (row < rows && k + j < K) ? p[j] : 0.f;
        r[i] = make_float4(t[0], t[1], t[2], t[3]);
      }
    } else {
      const int k = k0 + (tid >> 5) + 8 * i, row = row0 + (tid & 31) * 4;
      const float* p = base + (int64_t)k * ld + row;
      if (vec_ok && k < K && row + 3 < rows) {
        r[i] = *reinterpret_cast<const float4*>(p);
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (k < K && row + j < rows) ? p[j] : 0.f;
        r[i] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  }
}

template <bool KC, int BK>
__device__ __forceinline__ void slab_store(float* __restrict__ img, const float4 (&r)[nv_of(BK)]) {
  constexpr int LD = ld_of(KC), NV = nv_of(BK);
  constexpr int TPR = BK / 4, RPP = 256 / TPR;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (KC) {
      const int m = (tid / TPR) + RPP * i, k = (tid % TPR) * 4;
      img[(k + 0) * LD + m] = r[i].x;
      img[(k + 1) * LD + m] = r[i].y;
      img[(k + 2) * LD + m] = r[i].z;
      img[(k + 3) * LD + m] = r[i].w;
    } else {
      const int k = (tid >> 5) + 8 * i, m = (tid & 31) * 4;
      *reinterpret_cast<float4*>(img + k * LD + m) = r[i];
    }
  }
}

// ---- low-rank bias (GlobalBias with projection) accumulated straight into the MFMA accumulators:
//   acc[m,n] += sum_c pwT[c,m] * m8[c,n].  pwT is the transposed projection weight so that the four
//   consecutive rows (r&3) of an accumulator group come from one 16-byte load (needs M % 4 == 0).
//   One channel at a time keeps the live set at acc + 8 registers.
__device__ __forceinline__ void gemm_add_projection(const GemmArgs& g, f32x16 (&acc)[2][2], int m0, int n0,
                                                    int wm, int wn, int li, int lh) {
  const int nc0 = min(n0 + wn * 64 + li, g.N - 1), nc1 = min(n0 + wn * 64 + 32 + li, g.N - 1);
#pragma unroll 1
  for (int c = 0; c < g.cin; ++c) {
    const float mb0 = g.m8[(int64_t)c * g.N + nc0], mb1 = g.m8[(int64_t)c * g.N + nc1];
    const float* pc = g.pw + (int64_t)c * g.M;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int mrow = m0 + wm * 64 + tm * 32 + 4 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mr = mrow + 8 * j;
        const float4 p4 = (mr < g.M) ? *reinterpret_cast<const float4*>(pc + mr) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[tm][0][4 * j + 0] += p4.x * mb0; acc[tm][1][4 * j + 0] += p4.x * mb1;
        acc[tm][0][4 * j + 1] += p4.y * mb0; acc[tm][1][4 * j + 1] += p4.y * mb1;
        acc[tm][0][4 * j + 2] += p4.z * mb0; acc[tm][1][4 * j + 2] += p4.z * mb1;
        acc[tm][0][4 * j + 3] += p4.w * mb0; acc[tm][1][4 * j + 3] += p4.w * mb1;
      }
    }
  }
}

__device__ __forceinline__ float gate_sigmoid(float a) { return 1.0f / (1.0f + expf(-a)); }
// value of x rounded to bf16 (round to nearest even; a NaN stays a NaN: v_cvt_pk_bf16_f32)
__device__ __forceinline__ float round_bf16(float x) { return (float)(__bf16)x; }
// Activations of the bf16-mixed epilogue (R16): the value is rounded to bf16 - 8 significant bits - in the next instruction, so
// the hardware's 1-ulp exp2 / reciprocal stand in for expf and the IEEE division of act_apply / act_grad (common.h): about 9
// instead of about 29 vector instructions per element in an epilogue that was bound by exactly those (sixteen waves x 64 elements
// per lane and tensor).  SiLU only; GELU keeps the library functions.  The fp32-width schemes never call these.
__device__ __forceinline__ float sigmoid_r16(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * z));
}
__device__ __forceinline__ float act_apply_r16(float z, int act) {
  return act == PARADIS_ACT_SILU ? z * sigmoid_r16(z) : act_apply(z, act);
}
__device__ __forceinline__ float act_grad_r16(float z, int act) {
  if (act == PARADIS_ACT_SILU) {
    const float s = sigmoid_r16(z);
    return s * (1.0f + z * (1.0f - s));
  }
  return act_grad(z, act);
}
// (An epilogue / k-loop stagger - the second workgroup of every CU of the first round starting late by 64-256 x 512 cycles,
//  so that one workgroup's store-bound epilogue runs under the other's MFMA-bound k-loop - was measured on the bf16-mixed
//  and the bf16x3 kernels and lost 0-10 % at every setting: profiles/r06_stagger_sweep.txt.  Not kept.)
[[maybe_unused]] constexpr int IO_B16 = 1, IO_C16 = 2, IO_ZM16 = 4, IO_A16 = 8;     // GemmArgs::io16 (IO_A16: wgrad's dY operand)
// bf16 storage: element i of a bf16 array as a float / a bf16-VALUED float (already rounded) into a bf16 array
__device__ __forceinline__ float ld_bf16(const void* p, int64_t i) {
  return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16);
}
__device__ __forceinline__ void st_bf16(void* p, int64_t i, float v) {
  reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(__float_as_uint(v) >> 16);
}

// ---- epilogue: C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
//   v = acc (+bias[m]) (+map[m,n]); zout = v; v = zmul ? v*act'(zmul) : act(v);
//   v = gate ? res + sigmoid(gate[m]) (v - res) : v + res; C = v
// Interior tiles take a path without per-element guards in which all loads of one 32-row group are
// issued back to back (the guarded form serialises every load behind an s_waitcnt vmcnt(0)).
// R16 (compile time: only the bf16-mixed kernel instantiates it, the fp32 schemes' epilogue is the round-4 code): round
// the pre-activation and the activated / activation-gradient value to bf16 (see the note in GemmArgs)
template <bool R16 = false, bool C16 = false, bool ZM16 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], int bz, int m0,
                                              int n0, int wm, int wn, int li, int lh) {
  float* Cb = g.C + (int64_t)bz * g.c_bs;
  const float* resb = g.res ? g.res + (int64_t)bz * g.res_bs : nullptr;
  const float* zmulb = g.zmul ? g.zmul + (int64_t)bz * g.zmul_bs : nullptr;
  float* zoutb = g.zout ? g.zout + (int64_t)bz * g.zout_bs : nullptr;
  // bf16-stored tensors (compile-time properties of the bf16-mixed kernels' instantiations - as run-time branches they
  // cost the 128-register kernels 256 bytes of scratch and 100 us per launch): the same element offsets on 2-byte elements
  static_assert(R16 || !(C16 || ZM16), "bf16-stored tensors exist in the bf16-mixed scheme only");
  (void)sizeof(char[C16 + ZM16 + 1]);
  const int64_t cb16 = (int64_t)bz * g.c_bs, zmb16 = (int64_t)bz * g.zmul_bs, zob16 = (int64_t)bz * g.zout_bs;
  if (g.pw) gemm_add_projection(g, acc, m0, n0, wm, wn, li, lh);
  if constexpr (C16 || ZM16) {
    // bf16-stored tensors of an interior tile move as PACKED PAIRS: a lane holds pixel li of the wave's two 32-column MFMA
    // tiles (columns li and 32 + li of one row) - as 2-byte accesses a row of a tile is a 64-byte segment per instruction.
    // Adjacent lanes swap one value each (even lane: its tile-1 value for the odd lane's tile-0 value), after which the even
    // lane holds columns (li, li + 1) of tile 0 and the odd lane columns (31 + li, 32 + li): one dword per lane, 128 contiguous
    // bytes per row and instruction, half the instructions.  Loads of a bf16 zmul run the same exchange backwards.
    if (m0 + BM <= g.M && n0 + BN <= g.N && ((g.ldc | g.c_bs | g.zout_bs | g.zmul_bs) & 1) == 0) {
      const bool odd = (li & 1) != 0;
      const uint32_t sel = odd ? 0x07060302u : 0x03020706u;       // v_perm_b32(keep, recv): {lo, hi} halves of the dword
      const int pcol = odd ? 31 + li : li;                         // first column of this lane's pair (even)
      auto swap1 = [](float v) __attribute__((always_inline)) {   // the neighbour's value (lanes 2u <-> 2u + 1)
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
      };
      auto pack = [&](float a0, float a1) __attribute__((always_inline)) {      // a0 / a1: this lane's tile-0 / tile-1 value (bf16-valued)
        const float keep = odd ? a1 : a0, recv = swap1(odd ? a0 : a1);
        return __builtin_amdgcn_perm(__float_as_uint(keep), __float_as_uint(recv), sel);
      };
      auto unpack = [&](uint32_t w, float& t0, float& t1) __attribute__((always_inline)) {
        const float wlo = __uint_as_float(w << 16), whi = __uint_as_float(w & 0xffff0000u);
        const float recv = swap1(odd ? wlo : whi);
        t0 = odd ? recv : wlo;
        t1 = odd ? whi : recv;
      };
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const int mrow = m0 + wm * 64 + tm * 32 + 4 * lh;
        const int64_t base = (int64_t)mrow * g.ldc + n0 + wn * 64 + li;          // tile 0; tile 1: + 32
        const int64_t pbase = (int64_t)mrow * g.ldc + n0 + wn * 64 + pcol;       // this lane's pair
#pragma unroll
        for (int h = 0; h < 4; ++h) {   // 4 accumulator registers of each tile at a time
          float v[2][4], t[2][4];
          // register r = 4h + q  ->  row offset q + 8h
#define ROWOFF(q) ((int64_t)((q) + 8 * h) * g.ldc)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[tn][q] = acc[tm][tn][4 * h + q];
          if (g.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[0][q] = g.bias[mrow + q + 8 * h];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[0][q] += t[0][q]; v[1][q] += t[0][q]; }
          }
          if (g.map) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) t[tn][q] = g.map[base + 32 * tn + ROWOFF(q)];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[tn][q] += t[tn][q];
          }
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[tn][q] = round_bf16(v[tn][q]);
          if (zoutb) {
            if constexpr (C16) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                reinterpret_cast<uint32_t*>(g.zout)[(zob16 + pbase + ROWOFF(q)) >> 1] = pack(v[0][q], v[1][q]);
            } else {
#pragma unroll
              for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) zoutb[base + 32 * tn + ROWOFF(q)] = v[tn][q];
            }
          }
          if (zmulb) {
            if constexpr (ZM16) {
              uint32_t w[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) w[q] = reinterpret_cast<const uint32_t*>(g.zmul)[(zmb16 + pbase + ROWOFF(q)) >> 1];
#pragma unroll
              for (int q = 0; q < 4; ++q) unpack(w[q], t[0][q], t[1][q]);
            } else {
#pragma unroll
              for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) t[tn][q] = zmulb[base + 32 * tn + ROWOFF(q)];
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[tn][q] *= act_grad_r16(t[tn][q], g.act);
          } else if (g.act) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[tn][q] = act_apply_r16(v[tn][q], g.act);
          }
          if (zmulb || g.act) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[tn][q] = round_bf16(v[tn][q]);
          }
          if (resb) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) t[tn][q] = resb[base + 32 * tn + ROWOFF(q)];
            if (g.gate) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float gm = gate_sigmoid(g.gate[mrow + q + 8 * h]);
                v[0][q] = fmaf(gm, v[0][q] - t[0][q], t[0][q]);
                v[1][q] = fmaf(gm, v[1][q] - t[1][q], t[1][q]);
              }
            } else {
#pragma unroll
              for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[tn][q] += t[tn][q];
            }
          }
          if constexpr (C16) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              reinterpret_cast<uint32_t*>(g.C)[(cb16 + pbase + ROWOFF(q)) >> 1] = pack(v[0][q], v[1][q]);
          } else {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int q = 0; q < 4; ++q) Cb[base + 32 * tn + ROWOFF(q)] = v[tn][q];
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#undef ROWOFF
        }
      }
      return;
    }
  }
  if (m0 + BM <= g.M && n0 + BN <= g.N) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int mrow = m0 + wm * 64 + tm * 32 + 4 * lh;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int64_t base = (int64_t)mrow * g.ldc + n0 + wn * 64 + tn * 32 + li;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 8 accumulator registers at a time keeps the kernel <= 128 VGPRs
          float v[8], t[8];
          // register r = 8h + q  ->  row offset (q&3) + 8*(2h + (q>>2))
#define ROWOFF(q) ((int64_t)(((q) & 3) + 8 * (2 * h + ((q) >> 2))) * g.ldc)
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = acc[tm][tn][8 * h + q];
          if (g.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = g.bias[mrow + (q & 3) + 8 * (2 * h + (q >> 2))];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
          }
          if (g.map) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = g.map[base + ROWOFF(q)];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
          }
          if constexpr (R16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = round_bf16(v[q]);
          }
          if (zoutb) {
            if constexpr (C16) {
#pragma unroll
              for (int q = 0; q < 8; ++q) st_bf16(g.zout, zob16 + base + ROWOFF(q), v[q]);
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) zoutb[base + ROWOFF(q)] = v[q];
            }
          }
          if (zmulb) {
            if constexpr (ZM16) {
#pragma unroll
              for (int q = 0; q < 8; ++q) t[q] = ld_bf16(g.zmul, zmb16 + base + ROWOFF(q));
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) t[q] = zmulb[base + ROWOFF(q)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= R16 ? act_grad_r16(t[q], g.act) : act_grad(t[q], g.act);
          } else if (g.act) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = R16 ? act_apply_r16(v[q], g.act) : act_apply(v[q], g.act);
          }
          if constexpr (R16) {
            if (zmulb || g.act) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = round_bf16(v[q]);
            }
          }
          if (resb) {
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = resb[base + ROWOFF(q)];
            if (g.gate) {   // h + sigmoid(alpha) (adv - h): the arithmetic of gated_blend_fwd_kernel (misc.hip), bit for bit
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const float gm = gate_sigmoid(g.gate[mrow + (q & 3) + 8 * (2 * h + (q >> 2))]);
                v[q] = fmaf(gm, v[q] - t[q], t[q]);
              }
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] += t[q];
            }
          }
          if constexpr (C16) {
#pragma unroll
            for (int q = 0; q < 8; ++q) st_bf16(g.C, cb16 + base + ROWOFF(q), v[q]);
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) Cb[base + ROWOFF(q)] = v[q];
          }
          // keep the scheduler from hoisting the next chunk's loads (register pressure)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#undef ROWOFF
        }
      }
    }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= g.M) continue;
      const float bv = g.bias ? g.bias[m] : 0.f;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + li;
        if (n >= g.N) continue;
        const int64_t off = (int64_t)m * g.ldc + n;
        float v = acc[tm][tn][r] + bv;
        if (g.map) v += g.map[off];
        if constexpr (R16) v = round_bf16(v);
        if (zoutb) { if constexpr (C16) st_bf16(g.zout, zob16 + off, v); else zoutb[off] = v; }
        if (zmulb) {
          const float zm = ZM16 ? ld_bf16(g.zmul, zmb16 + off) : zmulb[off];
          v *= R16 ? act_grad_r16(zm, g.act) : act_grad(zm, g.act);
        } else if (g.act) {
          v = R16 ? act_apply_r16(v, g.act) : act_apply(v, g.act);
        }
        if constexpr (R16) { if (zmulb || g.act) v = round_bf16(v); }
        if (resb) {
          const float r = resb[off];
          v = g.gate ? fmaf(gate_sigmoid(g.gate[m]), v - r, r) : v + r;
        }
        if constexpr (C16) st_bf16(g.C, cb16 + off, v); else Cb[off] = v;
      }
    }
  }
}

template <bool A_KC, bool B_KC, int BK>
__global__ void __launch_bounds__(256, (BK == 16 ? 4 : 2))
pw_gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [stage][A|B][STAGE_FLOATS]
  constexpr int LDA = ld_of(A_KC), LDB = ld_of(B_KC);
  constexpr int NV = nv_of(BK), STAGE_FLOATS = stage_floats(BK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // ---- XCD-aware decode: consecutive logical ids (same X tile, different M tiles) share an XCD
  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;

  // k-tile range of this workgroup.  fwd/dgrad: all KT tiles of sample bz.  wgrad: the flattened
  // (sample, k-tile) sequence of inner*KT tiles is cut into nbatch equal contiguous ranges.
  const int KT = (g.K + BK - 1) / BK;
  int t_begin = 0, T = KT;
  if (g.inner > 0) {
    const int64_t total = (int64_t)g.inner * KT;
    t_begin = (int)(total * bz / g.nbatch);
    T = (int)(total * (bz + 1) / g.nbatch) - t_begin;
  }
  const float* Ab = g.A + (g.inner > 0 ? 0 : (int64_t)bz * g.a_bs);
  const float* Bb = g.B + (g.inner > 0 ? 0 : (int64_t)bz * g.b_bs);

  const bool a_vec = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                     ((g.a_bs & 3) == 0) && ((g.a_is & 3) == 0);
  const bool b_vec = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) &&
                     ((g.b_bs & 3) == 0) && ((g.b_is & 3) == 0);

  // Co-resident workgroups of one CU (dispatch ids 256 apart) run the same program with one
  // barrier per k-tile and drift into lockstep: their non-MFMA phases (LDS store, barrier, global
  // issue) then coincide and the matrix pipe idles.  Skew their start by a fraction of a k-tile.
  if (g.stagger > 0) {
    const int lag = (blockIdx.x >> 8) & 3;
    for (int i = 0; i < lag * g.stagger; ++i) __builtin_amdgcn_s_sleep(8);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NV], rb[NV];
  auto fetch = [&](int t) {
    const int tt = t_begin + t;
    const int ib = tt / KT, kt = tt - ib * KT;
    const float* Ap = Ab + (g.inner > 0 ? (int64_t)ib * g.a_is : 0);
    const float* Bp = Bb + (g.inner > 0 ? (int64_t)ib * g.b_is : 0);
    slab_load<A_KC, BK>(Ap, g.lda, m0, kt * BK, g.M, g.K, a_vec, ra);
    slab_load<B_KC, BK>(Bp, g.ldb, n0, kt * BK, g.N, g.K, b_vec, rb);
  };
  auto stageA = [&](int st) { return lds + (st * 2 + 0) * STAGE_FLOATS; };
  auto stageB = [&](int st) { return lds + (st * 2 + 1) * STAGE_FLOATS; };

  if (T > 0) {
    fetch(0);
    slab_store<A_KC, BK>(stageA(0), ra);
    slab_store<B_KC, BK>(stageB(0), rb);
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    if (t + 1 < T) fetch(t + 1);
    const float* As = stageA(cur) + wm * 64 + li + lh * LDA;
    const float* Bs = stageB(cur) + wn * 64 + li + lh * LDB;
    float a0 = As[0], a1 = As[32], b0 = Bs[0], b1 = Bs[32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (kk + 1 < BK / 2) {
        na0 = As[(2 * kk + 2) * LDA]; na1 = As[(2 * kk + 2) * LDA + 32];
        nb0 = Bs[(2 * kk + 2) * LDB]; nb1 = Bs[(2 * kk + 2) * LDB + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (t + 1 < T) {
      slab_store<A_KC, BK>(stageA(cur ^ 1), ra);
      slab_store<B_KC, BK>(stageB(cur ^ 1), rb);
    }
    __syncthreads();
  }

  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// ======================================================================================
// LDS-DMA variant for row-contiguous operands (A(m,k) at A[k*lda+m], B(k,n) at B[k*ldb+n]):
// fwd with pre-transposed weights, dgrad.  Tiles go global -> LDS with global_load_lds_dwordx4
// (no VGPR staging, no ds_write), an S-deep LDS ring, counted vmcnt and a raw barrier per k-tile,
// so that S-1 tiles of loads stay in flight behind the MFMAs (measured on the register-staged
// kernel: exposed global-load latency costs ~25 % of the matrix pipe).
// Requirements (checked by the host, else the register-staged kernel is used):
//   K % 16 == 0, lda/ldb/batch strides multiples of 4 floats, 16-B aligned bases, M % 4 == N % 4 == 0.
// Out-of-range rows/cols of edge tiles are clamped to valid addresses; they only feed outputs
// that the epilogue discards.
// ======================================================================================
constexpr int DBK = 16;                 // k-tile depth of the DMA kernel
constexpr int DTILE = DBK * BM;         // floats per operand per stage (pitch 128, unpadded)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int S, int MINW>
__global__ void __launch_bounds__(256, MINW)
pw_gemm_dma_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A|B][DTILE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = g.K / DBK;

  // this lane's source column inside a 2-row piece, clamped to stay inside the matrix
  const int pr = lane >> 5, pc = (lane & 31) * 4;
  const int acol = min(m0 + pc, g.M - 4), bcol = min(n0 + pc, g.N - 4);
  const float* Ap = g.A + (int64_t)bz * g.a_bs + (int64_t)(2 * (2 * wave) + pr) * g.lda + acol;
  const float* Bp = g.B + (int64_t)bz * g.b_bs + (int64_t)(2 * (2 * wave) + pr) * g.ldb + bcol;
  const int64_t a_piece = 2 * g.lda, b_piece = 2 * g.ldb;      // next 2-row piece
  const int64_t a_tile = (int64_t)DBK * g.lda, b_tile = (int64_t)DBK * g.ldb;

  auto issue = [&](int t) {
    float* st = lds + (t % S) * (2 * DTILE);
    const float* a = Ap + (int64_t)t * a_tile;
    const float* b = Bp + (int64_t)t * b_tile;
    // wave w owns pieces 2w, 2w+1 (k-rows 4w..4w+3) of both operands
    float* la = st + (2 * wave) * 256;
    float* lb = st + DTILE + (2 * wave) * 256;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)a, (lds_ptr_t)la, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a + a_piece), (lds_ptr_t)(la + 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)b, (lds_ptr_t)lb, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b + b_piece), (lds_ptr_t)(lb + 256), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < T) issue(t);

  for (int t = 0; t < T; ++t) {
    // tile t must have landed; up to S-2 younger tiles (4 DMAs each per wave) stay in flight
    const int pending = min(S - 2, T - 1 - t);
    if (pending >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (pending == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // the stage of tile t-1 is free now (every wave has passed its MFMAs): refill it
    if (t + S - 1 < T) issue(t + S - 1);
    const float* As = lds + (t % S) * (2 * DTILE) + wm * 64 + li + lh * BM;
    const float* Bs = As - wm * 64 + DTILE + wn * 64;
    float a0 = As[0], a1 = As[32], b0 = Bs[0], b1 = Bs[32];
#pragma unroll
    for (int kk = 0; kk < DBK / 2; ++kk) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (kk + 1 < DBK / 2) {
        na0 = As[(2 * kk + 2) * BM]; na1 = As[(2 * kk + 2) * BM + 32];
        nb0 = Bs[(2 * kk + 2) * BN]; nb1 = Bs[(2 * kk + 2) * BN + 32];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  }
  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// ======================================================================================
// LDS-DMA weight-gradient kernel: dW[M,N'] = sum over (sample, p) of A[m][p] * B[n][p] with BOTH
// operands k(=p)-contiguous.  A 128x16 slab is DMA'd as 8 pieces of 16 rows x 64 B into a
// [row][16 k] LDS image whose 16-B chunks are XOR-swizzled on the SOURCE side
// (slot = chunk ^ ((row>>2)&3)) so that ds_read_b128 of one chunk per lane is bank-conflict free.
// MFMA k-permutation: lanes 0-31 read chunk 2g, lanes 32-63 chunk 2g+1 of their row; MFMA e of
// group g then contracts k = {8g+e, 8g+4+e}; both operands use the same convention, so any
// permutation of k is legal.  8 ds_read_b128 per wave per k-tile instead of 32 ds_read_b32.
// ======================================================================================
template <int S>
__global__ void __launch_bounds__(256, 4)
pw_gemm_wgrad_dma_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A|B][128*16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = g.K / DBK;
  const int64_t total = (int64_t)g.inner * KT;
  const int t_begin = (int)(total * bz / g.nbatch);
  const int T = (int)(total * (bz + 1) / g.nbatch) - t_begin;

  // DMA lane mapping inside a 16-row piece: row = lane>>2, LDS slot = lane&3, source chunk swizzled
  const int prow = lane >> 2, pslot = lane & 3;
  const int chunk = pslot ^ ((prow >> 2) & 3);
  const int ra0 = min(m0 + 16 * (2 * wave) + prow, g.M - 1), ra1 = min(m0 + 16 * (2 * wave + 1) + prow, g.M - 1);
  const int rb0 = min(n0 + 16 * (2 * wave) + prow, g.N - 1), rb1 = min(n0 + 16 * (2 * wave + 1) + prow, g.N - 1);
  const float* pa0 = g.A + (int64_t)ra0 * g.lda + 4 * chunk;
  const float* pa1 = g.A + (int64_t)ra1 * g.lda + 4 * chunk;
  const float* pb0 = g.B + (int64_t)rb0 * g.ldb + 4 * chunk;
  const float* pb1 = g.B + (int64_t)rb1 * g.ldb + 4 * chunk;

  auto issue = [&](int t) {
    const int tt = t_begin + t;
    const int ib = tt / KT, kt = tt - ib * KT;
    const int64_t oa = (int64_t)ib * g.a_is + (int64_t)kt * DBK, ob = (int64_t)ib * g.b_is + (int64_t)kt * DBK;
    float* st = lds + (t % S) * (2 * DTILE);
    float* la = st + (2 * wave) * 256;
    float* lb = st + DTILE + (2 * wave) * 256;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pa0 + oa), (lds_ptr_t)la, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pa1 + oa), (lds_ptr_t)(la + 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pb0 + ob), (lds_ptr_t)lb, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(pb1 + ob), (lds_ptr_t)(lb + 256), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int t = 0; t < S - 1; ++t)
    if (t < T) issue(t);

  // bias gradient for free: the n-tile-0 workgroups also sum their A rows (dZ) over k.
  // thread t covers row t>>1, 16-B slots 2*(t&1), 2*(t&1)+1 (any chunk order: it is a plain sum)
  const bool do_rowsum = g.rowsum != nullptr && nt == 0;
  float rs = 0.f;
  const int rs_off = (tid >> 1) * DBK + (tid & 1) * 8;

  // fragment addressing: row r = w?*64 + t?*32 + li, slot = (2g+lh) ^ ((li>>2)&3)
  const int sw = (li >> 2) & 3;
  const int offA = (wm * 64 + li) * DBK, offB = DTILE + (wn * 64 + li) * DBK;
  const int s0 = ((0 + lh) ^ sw) * 4, s1 = ((2 + lh) ^ sw) * 4;

  for (int t = 0; t < T; ++t) {
    const int pending = min(S - 2, T - 1 - t);
    if (pending >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (pending == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + S - 1 < T) issue(t + S - 1);
    const float* st = lds + (t % S) * (2 * DTILE);
    if (do_rowsum) {
      const float4 q0 = *reinterpret_cast<const float4*>(st + rs_off);
      const float4 q1 = *reinterpret_cast<const float4*>(st + rs_off + 4);
      rs += ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
    }
    const float4 a00 = *reinterpret_cast<const float4*>(st + offA + s0);
    const float4 a10 = *reinterpret_cast<const float4*>(st + offA + 32 * DBK + s0);
    const float4 b00 = *reinterpret_cast<const float4*>(st + offB + s0);
    const float4 b10 = *reinterpret_cast<const float4*>(st + offB + 32 * DBK + s0);
    const float4 a01 = *reinterpret_cast<const float4*>(st + offA + s1);
    const float4 a11 = *reinterpret_cast<const float4*>(st + offA + 32 * DBK + s1);
    const float4 b01 = *reinterpret_cast<const float4*>(st + offB + s1);
    const float4 b11 = *reinterpret_cast<const float4*>(st + offB + 32 * DBK + s1);
#define MFMA4(A0, A1, B0, B1)                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);   \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);   \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);   \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);
    MFMA4(a00.x, a10.x, b00.x, b10.x)
    MFMA4(a00.y, a10.y, b00.y, b10.y)
    MFMA4(a00.z, a10.z, b00.z, b10.z)
    MFMA4(a00.w, a10.w, b00.w, b10.w)
    MFMA4(a01.x, a11.x, b01.x, b11.x)
    MFMA4(a01.y, a11.y, b01.y, b11.y)
    MFMA4(a01.z, a11.z, b01.z, b11.z)
    MFMA4(a01.w, a11.w, b01.w, b11.w)
#undef MFMA4
  }
  if (do_rowsum) {
    rs += __shfl_xor(rs, 1, 64);
    const int m = m0 + (tid >> 1);
    if ((tid & 1) == 0 && m < g.M) g.rowsum[(int64_t)bz * g.M + m] = rs;
  }
  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// ======================================================================================
// Split-bf16 kernels: the same fp32 GEMMs on the bf16 matrix pipe (16x the f32 MFMA rate).
// Every fp32 operand value x is written as h + m + l with h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m): three bf16 numbers carry 3 x 8 = 24 significand bits, so the split is exact.
// a*b is accumulated in fp32 from the six partial products of weight >= 2^-16,
//     ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm        (dropped: am*bl, al*bm, al*bl <= 2^-23 |ab|)
// v_mfma_f32_32x32x16_bf16 forms the 8x8-bit products exactly and accumulates in fp32, 6/16 roundings
// per k instead of the f32 MFMA's 1: the measured error against fp64 is below the f32 kernels'
// (tests/test_hip_gemm_split.py).  Non-finite inputs come out as NaN (Inf - Inf in the split).
//
// LDS image of a 128 x 16 operand tile: [split 3][k-half 2][row 128] chunks of 16 B = 8 bf16
// (k = 8*half + 0..7), so that one ds_read_b128 per lane (row = lane&31, half = lane>>5) is the
// MFMA operand.  Weights are split once per call into that image in global memory
// (split_weights_kernel) and move by LDS-DMA; activations are split in registers while staged.
// 2 stages x 24 KiB => 3 workgroups per CU.
// ======================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// LDS images are accessed through a clang vector type, not HIP's u32x4 struct: behind a struct-typed
// ds_read the compiler inserts an s_waitcnt vmcnt for every LDS-DMA still in flight (alias rule),
// which would serialise the DMA rings; vector-typed reads do not get that wait.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 16;                  // k depth of a tile = one bf16 / f16 MFMA
constexpr int SCH = 128;                 // chunks per k-half row of an unpadded image
constexpr int SCHP = 128 + 8;            // padded variant (wgrad: lane pairs write both k-halves of a row)
// NP = number of planes of an operand image = terms of the split: 3 = bf16 h/m/l (exact, six products),
// 2 = f16 h/l of the scaled value (22 significand bits, three products), 1 = the value rounded to bf16 (ONE product:
// PARADIS_GEMM_BF16, the arithmetic of the reference's bf16-mixed training mode - train.py:56 - never the fp32 path's)
constexpr int simg(int np) { return np * 2 * SCH; }      // chunks per operand per stage (12 / 8 KiB)
constexpr int simgp(int np) { return np * 2 * SCHP; }
constexpr int SIMG = simg(3);

// ---- sign checkerboard (round 4) ---------------------------------------------------------------------------------
// What the bf16 / f16 MFMA does with its accumulator input (tools/mfma_round_probe.hip, gfx950): the sixteen products
// are summed first; the smaller of {C, product sum} is then aligned to the larger one's exponent, and when the smaller
// one is C its low bits are dropped by a two's-complement FLOOR (C = -2^-26 against a product sum of 1 - 1 comes out as
// -2^-24).  Whenever a k-tile's product sum outweighs the running accumulator - the first tiles, and every later zero
// crossing: ~5 times per K = 1024 dot product of zero-mean data - the result moves half a granule towards -infinity.
// The f32 MFMA (an fma chain, round to nearest) has no such term.  Measured (tools/gemm_bias_check.py, N(0,1) data,
// unit u = 2^-24 rms(C)): every output of the six-product GEMM carried the SAME offset, -0.9 u on top of 8.2 u of
// zero-mean noise (f32 MFMA: +0.002 u on 9.6 u), -8 u on the weight gradient's 32,768-term sums.  Harmless per element,
// but sums over pixels or channels of a GEMM output (bias / ChannelNorm parameter gradients over 32,768 points, the
// per-pixel channel statistics) add the offset coherently where noise averages out: 0.9 u x 32,768 against 8.2 u x 181.
// Remedy without a second accumulator set or VALU work per tile: run alternate 32-row x 64-column blocks of the output
// in the NEGATED space.  The weight image holds the rows of odd 32-row blocks with the opposite sign (free: written once
// by split_weights_kernel), the activation columns of odd 64-column blocks - the columns ONE wave stages - are negated
// while they are split in registers (-x splits exactly into -h, -m, -l; in the 128 x 256 kernel the sign is a
// compile-time property of the code path a staging wave takes, so it rides on source modifiers), so block (tm) of
// compute wave (wm, wn) accumulates (-1)^(tm+wn) C: there the floor acts on -C, the offset of C is +0.9 u, and the
// epilogue flips those blocks back.  The offset is still there per element; it alternates in sign every 32 rows and
// 64 columns and cancels in every sum over more than a block.  The weight gradient alternates by K-range slab instead (odd slabs negate dY): there the offsets of an
// element's slabs cancel in the slab sum.
#ifndef SPLIT_STAGGER         // (-DSPLIT_STAGGER=1: A/B build of the staggered 128 x 256 kernel)
#define SPLIT_STAGGER 0
#endif
#ifndef SPLIT_SIGNED          // (-DSPLIT_SIGNED=0: the unsigned accumulation of rounds 1-3, for A/B runs)
#define SPLIT_SIGNED 1
#endif
#ifndef SPLIT_SIGNED_WGRAD    // (the slab alternation of the weight gradient alone)
#define SPLIT_SIGNED_WGRAD SPLIT_SIGNED
#endif
// sign bit of the staging thread's activation column (column = tid & 127 of a 128-column tile)
__device__ __forceinline__ uint32_t split_flip_mask(int col) { return SPLIT_SIGNED && (col & 64) ? 0x80000000u : 0u; }
__device__ __forceinline__ void flip8(float (&y)[8], const float (&x)[8], uint32_t mask) {
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = __uint_as_float(__float_as_uint(x[j]) ^ mask);
}
// the row blocks tm with (tm + wn) odd hold -C (wn = the wave's 64-column half: the sign of its activation columns)
__device__ __forceinline__ void split_unflip(f32x16 (&acc)[2][2], int wn) {
  if (!SPLIT_SIGNED) return;
  const float s0 = wn ? -1.f : 1.f, s1 = -s0;          // wave-uniform
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][tn][r] *= s0; acc[1][tn][r] *= s1; }
}

// ---- f16x2 scheme ------------------------------------------------------------------------------
// x' = x 2^e with e chosen per TENSOR so that max |x'| lies in [2^14, 2^15) (fp16 holds 65504);
// h = f16(x'), l = f16(x' - h): x' = h + l up to 2^-23 |x'|, and - fp16 being a fixed-point format below
// 2^-14 - up to 2^-25 absolutely, i.e. 2^-39 of the tensor's largest magnitude.  a b is accumulated in
// fp32 from  al bh + ah bl + ah bh  (dropped: al bl <= 2^-22 |ab|); the f16 MFMA forms the 11x11-bit
// products exactly.  The result is unscaled by 2^-(ea+eb) in the epilogue (two exact multiplications).
// Error against fp64 of a K = 1024 product of N(0,1) operands: 5.1e-7 of max |C| (SGEMM: 5.8e-7); what
// it gives up against the bf16x3 scheme is elements more than ~2^17 below their tensor's maximum, which
// keep an ABSOLUTE accuracy of 2^-39 max|x| instead of a relative one.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// scale 2^e and its inverse from the bits of max |x| (biased exponent E: e = 14 - (E - 127)); a zero or
// tiny maximum takes the largest scale.  A non-finite maximum (an Inf or NaN somewhere in the tensor) has no
// meaningful scale: the scale becomes NaN and with it the whole product - loudly wrong, never silently rescaled.
__device__ __forceinline__ void scale_from_amax(uint32_t amax_bits, float& s, float& inv) {
  int E = (int)((amax_bits >> 23) & 0xffu);
  const bool finite = E != 255;
  E = E < 15 ? 15 : E;
  s = finite ? __uint_as_float((uint32_t)(268 - E) << 23) : __uint_as_float(0x7fc00000u);
  inv = __uint_as_float((uint32_t)(E - 14) << 23);
}

// max of PARADIS_AMAX_PARTIALS (= 1024) words, by a 256-thread workgroup; every thread gets the result
__device__ __forceinline__ uint32_t reduce_amax_partials(const uint32_t* __restrict__ p) {
  __shared__ uint32_t red[4];
  const int tid = threadIdx.x;
  uint32_t m = max(max(p[tid], p[tid + 256]), max(p[tid + 512], p[tid + 768]));
  m = wave_umax_lane63(m);
  if ((tid & 63) == 63) red[tid >> 6] = m;
  __syncthreads();
  m = max(max(red[0], red[1]), max(red[2], red[3]));
  __syncthreads();
  return m;
}

// (x0, x1) 2^e -> packed halves h, l.  v_fma_mix*: fp32 FMA, result rounded once to f16; x s and
// x s - h are exact in fp32, so h and l are the correctly rounded values.
__device__ __forceinline__ void split2_pair(float x0, float x1, float s, uint32_t& h, uint32_t& l) {
  uint32_t hh, ll;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(ll) : "v"(x0), "v"(s), "v"(hh));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(ll) : "v"(x1), "v"(s), "v"(hh));
  h = hh; l = ll;
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32, a in the low half
}

__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = pack_bf16(r0, r1);
  l = pack_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
  uint32_t hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair(x[2 * i], x[2 * i + 1], hh[i], mm[i], ll[i]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  m = (u32x4){mm[0], mm[1], mm[2], mm[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

// eight values rounded to bf16 (the one plane of PARADIS_GEMM_BF16)
__device__ __forceinline__ u32x4 round8(const float (&x)[8]) {
  return (u32x4){pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7])};
}

__device__ __forceinline__ void split8_f16(const float (&x)[8], float s, u32x4& h, u32x4& l) {
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2_pair(x[2 * i], x[2 * i + 1], s, hh[i], ll[i]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

// max |x| over B blocks of `inner` contiguous floats (block stride bs) -> PARADIS_AMAX_PARTIALS words, one
// per workgroup (bits of a non-negative float order like unsigned integers; a NaN is larger than Inf and
// so survives).  The consumers take the maximum of the words: no atomics, no zero-fill, deterministic.
#if GEMM_PART < 2
__global__ void __launch_bounds__(256)
amax_partials_kernel(const float* __restrict__ x, int B, int64_t inner, int64_t bs, int vec, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  if (vec) {
    const int64_t n4 = inner >> 2, total = n4 * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t b = i / n4, j = i - b * n4;
      const uint4 q = *reinterpret_cast<const uint4*>(x + b * bs + 4 * j);
      m = max(max(m, q.x & 0x7fffffffu), max(q.y & 0x7fffffffu, max(q.z & 0x7fffffffu, q.w & 0x7fffffffu)));
    }
  } else {
    const int64_t total = inner * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t b = i / inner, j = i - b * inner;
      m = max(m, __float_as_uint(x[b * bs + j]) & 0x7fffffffu);
    }
  }
  m = wave_umax_lane63(m);
  __shared__ uint32_t red[4];
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = max(max(red[0], red[1]), max(red[2], red[3]));
}
#endif

// Image of A[m,k] = W[m*rs + k*cs] (rs/cs select W or W^T), zero padded to [MT*128, KT*16]:
// out[((mt*KT + kt)*3 + s)*256 + half*128 + row] ; one thread per (mt, kt, half, row).
template <int NP = 3>
__device__ __forceinline__ void split_weights_body(const float* __restrict__ Wb, int64_t rs, int64_t cs, int M, int K,
                                                   int KT, int64_t units, u32x4* __restrict__ ob) {
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
    const int row = (int)(u & 127), half = (int)((u >> 7) & 1);
    const int64_t tile = u >> 8;
    const int kt = (int)(tile % KT), mt = (int)(tile / KT);
    const int m = mt * BM + row;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt * SBK + half * 8 + j;
      x[j] = (m < M && k < K) ? Wb[(int64_t)m * rs + (int64_t)k * cs] : 0.f;
      if (SPLIT_SIGNED && (row & 32)) x[j] = -x[j];      // sign checkerboard: odd 32-row blocks hold -W
    }
    if constexpr (NP == 3) {
      u32x4 h, mm, l;
      split8(x, h, mm, l);
      u32x4* o = ob + tile * SIMG + half * SCH + row;
      o[0] = h; o[2 * SCH] = mm; o[4 * SCH] = l;
    } else {
      ob[tile * simg(1) + half * SCH + row] = round8(x);
    }
  }
}

template <int NP>
__global__ void __launch_bounds__(256)
split_weights_kernel(const float* __restrict__ W, int64_t rs, int64_t cs, int M, int K, int KT, int64_t units,
                     int64_t w_bs, int64_t out_bs, u32x4* __restrict__ out) {
  split_weights_body<NP>(W + (int64_t)blockIdx.y * w_bs, rs, cs, M, K, KT, units, out + (int64_t)blockIdx.y * out_bs);
}

// both images of one row-major W[M,K] in ONE launch (a training step needs W for the forward GEMM and W^T for the
// data gradient: 78 launches of a few microseconds per step instead of 155): blockIdx.y = 0 -> W, 1 -> W^T
template <int NP>
__global__ void __launch_bounds__(256)
split_weights_pair_kernel(const float* __restrict__ W, int M, int K, int KT, int KTt, int64_t units, int64_t units_t,
                          u32x4* __restrict__ out, u32x4* __restrict__ out_t) {
  if (blockIdx.y == 0) split_weights_body<NP>(W, K, 1, M, K, KT, units, out);
  else split_weights_body<NP>(W, 1, K, K, M, KTt, units_t, out_t);
}

// f16x2 image: out[((mt*KT + kt)*2 + s)*256 + half*128 + row]; `tail` = the words behind the image:
// [0] = bits of max |W| (written here, read by the GEMMs), [4 ..) = the amax partials of W (input)
#if GEMM_PART < 2
__global__ void __launch_bounds__(256)
split_weights_f16_kernel(const float* __restrict__ W, int64_t rs, int64_t cs, int M, int K, int KT, int64_t units,
                         u32x4* __restrict__ out, uint32_t* __restrict__ tail) {
  const uint32_t amax = reduce_amax_partials(tail + 4);
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[0] = amax;
  float sc, inv;
  scale_from_amax(amax, sc, inv);
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
    const int row = (int)(u & 127), half = (int)((u >> 7) & 1);
    const int64_t tile = u >> 8;
    const int kt = (int)(tile % KT), mt = (int)(tile / KT);
    const int m = mt * BM + row;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt * SBK + half * 8 + j;
      x[j] = (m < M && k < K) ? W[(int64_t)m * rs + (int64_t)k * cs] : 0.f;
    }
    u32x4 h, l;
    split8_f16(x, (SPLIT_SIGNED && (row & 32)) ? -sc : sc, h, l);      // sign checkerboard
    u32x4* o = out + tile * simg(2) + half * SCH + row;
    o[0] = h; o[2 * SCH] = l;
  }
}
#endif

#define SPLIT_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
// The six partial products of one 32x32 block, smallest first.  (An order in which consecutive MFMAs
// share an operand register, snaking over the four blocks of a wave tile, measured +0.5 % - nothing -
// once the A/B alternated the variants: a fixed order of variants shows 3-5 % in favour of whichever
// runs later.)
#define SPLIT_BLOCK(AH, AM, AL, BH, BM_, BL, C) \
  SPLIT_MFMA(AM, BM_, C); SPLIT_MFMA(AL, BH, C); SPLIT_MFMA(AH, BL, C); \
  SPLIT_MFMA(AM, BH, C);  SPLIT_MFMA(AH, BM_, C); SPLIT_MFMA(AH, BH, C)
#define SPLIT_MFMA16(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// one k-tile: fragments of both operands from the images at As/Bs (chunk pointers at this lane's
// row of block 0, k-half lh), plane stride PA/PB chunks ...
template <int NP> struct SplitFrags { u32x4 a[NP][2], b[NP][2]; };
template <int NP, int PA, int PB>
__device__ __forceinline__ void split_tile_read(const u32x4* As, const u32x4* Bs, SplitFrags<NP>& f) {
  // in the order of first use by split_tile_mfma (block (0,0): m.m, l.h, h.l first; l.h, h.l for two
  // planes), so that the counted lgkmcnt waits let the first MFMAs start after two reads
  if constexpr (NP == 3) {
    f.a[1][0] = As[PA];          f.b[1][0] = Bs[PB];
    f.a[2][0] = As[2 * PA];      f.b[0][0] = Bs[0];
    f.a[0][0] = As[0];           f.b[2][0] = Bs[2 * PB];
    f.b[1][1] = Bs[PB + 32];     f.b[0][1] = Bs[32];          f.b[2][1] = Bs[2 * PB + 32];
    f.a[1][1] = As[PA + 32];     f.a[2][1] = As[2 * PA + 32]; f.a[0][1] = As[32];
  } else if constexpr (NP == 2) {
    f.a[1][0] = As[PA];          f.b[0][0] = Bs[0];
    f.a[0][0] = As[0];           f.b[1][0] = Bs[PB];
    f.b[0][1] = Bs[32];          f.b[1][1] = Bs[PB + 32];
    f.a[1][1] = As[PA + 32];     f.a[0][1] = As[32];
  } else {
    f.a[0][0] = As[0];  f.b[0][0] = Bs[0];
    f.b[0][1] = Bs[32]; f.a[0][1] = As[32];
  }
}
// ... then the 24 (12) MFMAs, smallest products first
template <int NP>
__device__ __forceinline__ void split_tile_mfma(const SplitFrags<NP>& f, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      if constexpr (NP == 3) {
        SPLIT_BLOCK(f.a[0][tm], f.a[1][tm], f.a[2][tm], f.b[0][tn], f.b[1][tn], f.b[2][tn], acc[tm][tn]);
      } else if constexpr (NP == 2) {
        SPLIT_MFMA16(f.a[1][tm], f.b[0][tn], acc[tm][tn]);
        SPLIT_MFMA16(f.a[0][tm], f.b[1][tn], acc[tm][tn]);
        SPLIT_MFMA16(f.a[0][tm], f.b[0][tn], acc[tm][tn]);
      } else {
        SPLIT_MFMA(f.a[0][tm], f.b[0][tn], acc[tm][tn]);
      }
    }
}

// f16x2: C = 2^-(ea+eb) acc, two exact multiplications (their product may lie outside the fp32 range)
__device__ __forceinline__ void split_unscale(f32x16 (&acc)[2][2], float inv_a, float inv_b) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * inv_a) * inv_b;
}

// fwd / dgrad:  C_b = epi( A . B_b ),  A = split weight image (g.A, batch stride g.a_bs chunks),
// B_b[K,N] fp32 with n contiguous.
//
// Pipeline per k-tile t (one barrier per tile, two LDS stages):
//   fragment reads of t  ->  weight DMA of t+1, activation loads of t+2 (registers, two sets used
//   alternately: the loop is unrolled by two so that each set is a fixed register range)  ->
//   the 24 MFMAs of t with the bf16 split of t+1's activations interleaved between them
//   (sched_group_barrier: the VALU work issues in the shadow of the MFMAs of the same wave)  ->
//   ds_write of t+1  ->  s_waitcnt vmcnt(8): the DMA has landed, the loads of t+2 stay in flight.
//
// weight-image ring depth per scheme (stages; the DMA runs stages - 1 tiles ahead).  bf16x3: 2 (48 KiB,
// 3 WGs/CU; 4 stages = 72 KiB, 2 WGs/CU measured -2 %).  f16x2: its 8 KiB stages make a deeper ring free.
#ifndef SPLIT_ASTAGES_F16
#define SPLIT_ASTAGES_F16 2
#endif
constexpr int split_astages(int np) { return np == 2 ? SPLIT_ASTAGES_F16 : 2; }
template <int NP>
__global__ void __launch_bounds__(256, 3)
pw_gemm_split_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SIMG = simg(NP);                     // (shadows the bf16 constant)
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [2 activation stages][SIMG] | [SA weight stages][SIMG]
  constexpr int SA = split_astages(NP), DA = SA - 1;     // weight ring depth, DMA distance in tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = (g.K + SBK - 1) / SBK;

  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T * SIMG + tid;
  // k-half staged by this thread's wave (waves 0,1 -> 0; 2,3 -> 1): row addresses stay scalar
  const int bh = __builtin_amdgcn_readfirstlane(tid >> 7);
  // (uniform, but derived from integer divisions done on the vector unit: pin it to scalar registers)
  const float* Bb;
  {
    const uint64_t a = reinterpret_cast<uint64_t>(g.B + (int64_t)bz * g.b_bs);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    Bb = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
  }
  const int bn = min(n0 + (tid & 127), g.N - 1);

  float sc_b = 1.f, inv_a = 1.f, inv_b = 1.f;       // f16x2: activation scale, inverse scales of both operands
  if constexpr (NP == 2) {
    float sc_a;
    scale_from_amax(reduce_amax_partials(g.b_amax), sc_b, inv_b);
    scale_from_amax(g.a_amax[0], sc_a, inv_a);
  }

  const uint32_t flip = split_flip_mask(tid & 127);
  if constexpr (NP == 2) sc_b = __uint_as_float(__float_as_uint(sc_b) ^ flip);
  float xb[2][8] = {};     // defined values: the surplus split of the last tile reads a set that was never loaded
  auto issueA = [&](int t) __attribute__((always_inline)) {
    const u32x4* a = Ag + (int64_t)t * SIMG;
    u32x4* la = img + (2 + t % SA) * SIMG + wave * 64;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a + i * 256), (lds_ptr_t)(la + i * 256), 16, 0, 0);
  };
  auto split_store = [&](const float (&x)[8], u32x4* o) __attribute__((always_inline)) {
    if constexpr (NP == 3) {
      u32x4 h, m, l;
      float xs[8];
      flip8(xs, x, flip);          // sign checkerboard: odd 64-column blocks are staged negated
      split8(xs, h, m, l);
      o[0] = h; o[2 * SCH] = m; o[4 * SCH] = l;
    } else {
      u32x4 h, l;
      split8_f16(x, sc_b, h, l);   // (the column's sign rides on the scale)
      o[0] = h; o[2 * SCH] = l;
    }
  };
  // Activation loads are issued from inline asm (saddr form: scalar row base + 32-bit lane offset, no
  // vector address arithmetic) so that the compiler does not account for them: on this loop its own
  // bookkeeping degrades to s_waitcnt vmcnt(0) in front of the first use, which would also wait for the
  // loads of the tile after and for the weight DMA just issued.  The waits are counted by hand (use_x).
  // Rows beyond K re-read row K-1: they meet the zero padding of the weight image, and finite x 0 = 0
  // (a non-finite row K-1 poisons every output anyway), so no zero-fill is needed.
  const uint32_t boff = (uint32_t)bn * 4u;
  auto fetchB = [&](int t, float (&x)[8]) __attribute__((always_inline)) {
    const int k0 = t * SBK + bh * 8;
    const float* p = Bb + (int64_t)min(k0, g.K - 1) * g.ldb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("global_load_dword %0, %1, %2" : "=&v"(x[j]) : "v"(boff), "s"(p) : "memory");
      p += (k0 + j + 1 < g.K) ? g.ldb : 0;
    }
  };
  // Wait until at most N vector-memory operations issued after x's loads are outstanding.  x is an INPUT
  // of the asm (an in/out operand lets the compiler copy the not-yet-landed registers in front of the
  // wait), and a sched_barrier behind it keeps every read of x below.
#define USE_X(x, N) do { asm volatile("s_waitcnt vmcnt(" #N ")" :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), \
                                      "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]) : "memory");                  \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
  u32x4* const Bst = img + bh * SCH + (tid & 127);   // this thread's chunk in the activation image of stage 0

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue (once per 64-tile range: waited for in full)
  for (int u = 0; u < DA && u < T; ++u) issueA(u);
  fetchB(0, xb[0]);
  USE_X(xb[0], 0);
  if (T > 1) fetchB(1, xb[1]);
  split_store(xb[0], Bst);
  // raw barriers with counted waits: __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier and
  // would make every barrier wait for the activation loads that are meant to stay in flight
  if (T > 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  auto step = [&](int t, int cur, float (&xload)[8], float (&xsplit)[8]) __attribute__((always_inline)) {
    const u32x4* As = img + (2 + t % SA) * SIMG + lh * SCH + wm * 64 + li;
    const u32x4* Bs = img + cur * SIMG + lh * SCH + wn * 64 + li;
    const bool dmaA = t + DA < T, ldB = t + 2 < T;
    if (dmaA) issueA(t + DA);
    if (ldB) fetchB(t + 2, xload);
    // xsplit (tile t+1) was loaded a step ago; younger operations: this step's NP DMA and 8 loads
    if (dmaA && ldB) { if constexpr (NP == 3) USE_X(xsplit, 11); else USE_X(xsplit, 10); }
    else if (ldB) USE_X(xsplit, 8);
    else USE_X(xsplit, 0);
    // The fragment reads sit in the block of the MFMAs (behind the branches above the compiler's lgkmcnt
    // bookkeeping falls back to lgkmcnt(0) in front of the first MFMA; inside one block the waits are
    // counted and the first MFMA starts after two of the twelve reads).
    SplitFrags<NP> f;
    split_tile_read<NP, 2 * SCH, 2 * SCH>(As, Bs, f);
    // One basic block for every tile, the last included (its split writes a stage that nobody reads any
    // more): a second copy of the MFMA block behind a branch costs 32 accumulator moves per tile.
    split_tile_mfma<NP>(f, acc);
    split_store(xsplit, Bst + (cur ^ 1) * SIMG);
    // without this pinning, the training step 0.9 % slower (tools/ab_step.sh, same box, 3 of 3 rounds).
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);   // all fragment reads first, in first-use order
#pragma unroll
    for (int i = 0; i < (NP == 3 ? 24 : 12); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 3 : 2, 0);
    }
    // weight tile t+1 landed (its DMA is DA steps old: 8 loads of that step + 8 + NP operations per step since
    // are younger), own ds_writes done, the loads of t+2 and the younger DMAs still in flight
    if (dmaA && ldB) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(8 + (8 + NP) * (DA - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < T; t += 2) {
    step(t, 0, xb[0], xb[1]);
    if (t + 1 < T) step(t + 1, 1, xb[1], xb[0]);
  }
  split_unflip(acc, wn);
  if constexpr (NP == 2) split_unscale(acc, inv_a, inv_b);
  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// f16x2 forward / dgrad with a 128 x 256 workgroup tile: 8 waves = two 128-column halves (sub 0 / 1 = n-tiles
// 2 nt2, 2 nt2 + 1, each staging its own activation tile exactly like pw_gemm_split_kernel<2>) that share ONE
// weight tile and its DMA ring.  The k-loop of this GEMM is bound by the bytes it pulls out of L2 (DESIGN.md
// 4.1c): 24 KiB per two 128 x 128 x 16 tiles here instead of 32.  48 KiB of LDS, <= 128 VGPRs: two workgroups =
// 16 waves per CU.  An odd last n-tile leaves sub 1 without work: it runs along on the clamped last tile and
// skips the epilogue.
template <int NSUB, int NP = 2>
__global__ void __launch_bounds__(256 * NSUB, 4)      // (second argument: waves per SIMD)
pw_gemm_split_wide_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SIMG = simg(NP), SA = 2, DA = SA - 1;
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [NSUB][2 activation stages][SIMG] | [SA weight stages][SIMG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = __builtin_amdgcn_readfirstlane(wave >> 2), lw = wave & 3, ltid = tid & 255;
  const int wm = lw >> 1, wn = lw & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN, NT2 = (NT + NSUB - 1) / NSUB;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt2 = (L / MT) % NT2, bz = L / (MT * NT2);
  const bool live = NSUB * nt2 + sub < NT;           // wave-uniform
  const int nt = min(NSUB * nt2 + sub, NT - 1);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = (g.K + SBK - 1) / SBK;

  // the weight tile (SIMG = 256 NP chunks of 16 bytes) goes by LDS-DMA, one chunk per thread and piece: f16x2 one
  // piece of 512 chunks (the first 512 threads), bf16x3 a piece of 512 and a piece of 256 (waves 0-3)
  const bool doA = NSUB == 2 || wave < 8;            // wave-uniform
  const bool doA2 = NP == 3 && wave < 4;             // wave-uniform: second piece
  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T * SIMG + (tid & 511);
  const int bh = __builtin_amdgcn_readfirstlane(ltid >> 7);      // k-half staged by this wave
  const float* Bb;
  {
    const uint64_t a = reinterpret_cast<uint64_t>(g.B + (int64_t)bz * g.b_bs);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    Bb = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
  }
  const int bn = min(n0 + (ltid & 127), g.N - 1);

  float sc_b = 1.f, inv_a = 1.f, inv_b = 1.f;
  if constexpr (NP == 2) {
    float sc_a;
    __shared__ uint32_t red[4 * NSUB];
    const uint32_t* pp = g.b_amax;
    uint32_t m = NSUB == 2 ? max(pp[tid], pp[tid + 512]) : pp[tid];
    m = wave_umax_lane63(m);
    if (lane == 63) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 4 * NSUB; ++i) m = max(m, red[i]);
    scale_from_amax(m, sc_b, inv_b);
    scale_from_amax(g.a_amax[0], sc_a, inv_a);
  }

  const uint32_t flip = split_flip_mask(ltid & 127);
  if constexpr (NP == 2) sc_b = __uint_as_float(__float_as_uint(sc_b) ^ flip);
  float xb[2][8] = {};     // defined values: the surplus split of the last tile reads a set that was never loaded
  auto issueA = [&](int t) __attribute__((always_inline)) {
    if (doA)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t * SIMG), (lds_ptr_t)(img + (2 * NSUB + t % SA) * SIMG + wave * 64), 16, 0, 0);
    if (doA2)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t * SIMG + 512), (lds_ptr_t)(img + (2 * NSUB + t % SA) * SIMG + 512 + wave * 64), 16, 0, 0);
  };
  auto split_store = [&](const float (&x)[8], u32x4* o) __attribute__((always_inline)) {
    if constexpr (NP == 3) {
      u32x4 h, m, l;
      // sign checkerboard: odd 64-column blocks are staged negated.  (Two code paths behind a wave-uniform branch
      // with the sign folded into source modifiers - no v_xor - measured SLOWER, 154.8 against 153.0 ms per step: the
      // branch takes the split out of the MFMA block's schedule.)
      float xs[8];
      flip8(xs, x, flip);
      split8(xs, h, m, l);
      o[0] = h; o[2 * SCH] = m; o[4 * SCH] = l;
    } else {
      u32x4 h, l;
      split8_f16(x, sc_b, h, l);   // (the column's sign rides on the scale)
      o[0] = h; o[2 * SCH] = l;
    }
  };
  // inline-asm loads with hand-counted waits: see pw_gemm_split_kernel
  const uint32_t boff = (uint32_t)bn * 4u;
  auto fetchB = [&](int t, float (&x)[8]) __attribute__((always_inline)) {
    const int k0 = t * SBK + bh * 8;
    const float* p = Bb + (int64_t)min(k0, g.K - 1) * g.ldb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("global_load_dword %0, %1, %2" : "=&v"(x[j]) : "v"(boff), "s"(p) : "memory");
      p += (k0 + j + 1 < g.K) ? g.ldb : 0;
    }
  };
#define USE_X(x, N) do { asm volatile("s_waitcnt vmcnt(" #N ")" :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), \
                                      "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]) : "memory");                  \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
  u32x4* const Bst = img + sub * 2 * SIMG + bh * SCH + (ltid & 127);   // this thread's chunk in its sub's stage 0

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int u = 0; u < DA && u < T; ++u) issueA(u);
  fetchB(0, xb[0]);
  USE_X(xb[0], 0);
  if (T > 1) fetchB(1, xb[1]);
  split_store(xb[0], Bst);
  if (T > 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  auto step = [&](int t, int cur, float (&xload)[8], float (&xsplit)[8]) __attribute__((always_inline)) {
    const u32x4* As = img + (2 * NSUB + t % SA) * SIMG + lh * SCH + wm * 64 + li;
    const u32x4* Bs = img + (sub * 2 + cur) * SIMG + lh * SCH + wn * 64 + li;
    const bool dmaA = t + DA < T, ldB = t + 2 < T;
    if (dmaA) issueA(t + DA);
    if (ldB) fetchB(t + 2, xload);
    // xsplit (tile t+1) was loaded a step ago; younger operations: this step's DMA piece(s) and 8 loads
    if (dmaA && ldB && doA2) USE_X(xsplit, 10);
    else if (dmaA && ldB && doA) USE_X(xsplit, 9);
    else if (ldB) USE_X(xsplit, 8);
    else USE_X(xsplit, 0);
    if constexpr (NP == 3) {
      // three planes at 128 registers: the B fragments of ONE plane at a time (8 registers instead of 24), planes
      // in the order l, m, h so that the products still arrive roughly smallest first:
      //   ah.bl | am.bm, ah.bm | al.bh, am.bh, ah.bh
      auto mfma_block = [&]() __attribute__((always_inline)) {
        u32x4 a[3][2], b[2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { a[pl][0] = As[pl * 2 * SCH]; a[pl][1] = As[pl * 2 * SCH + 32]; }
#pragma unroll
        for (int pb = 2; pb >= 0; --pb) {
          b[0] = Bs[pb * 2 * SCH]; b[1] = Bs[pb * 2 * SCH + 32];
#pragma unroll
          for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
              for (int tn = 0; tn < 2; ++tn) SPLIT_MFMA(a[pa][tm], b[tn], acc[tm][tn]);
        }
      };
#if SPLIT_STAGGER
      // SIMD partners out of phase (MI355X_MICROARCH.md, two waves per SIMD, item 9): waves 4-7 (sub 1) split and store
      // tile t+1 FIRST and multiply afterwards, waves 0-3 the other way round - one half of a SIMD's waves is on the
      // vector unit and the LDS store path while the other half feeds the matrix pipe
      // (ONE copy of the MFMA block: a second copy behind the branch spills the accumulators)
      if (sub == 1) split_store(xsplit, Bst + (cur ^ 1) * SIMG);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block();
      __builtin_amdgcn_sched_barrier(0);
      if (sub == 0) split_store(xsplit, Bst + (cur ^ 1) * SIMG);
#else
      mfma_block();
      split_store(xsplit, Bst + (cur ^ 1) * SIMG);
#endif
      // (no sched_group_barrier pinning here: the 1 MFMA : 3 VALU pattern of the 128 x 128 kernel measured 0.7 % slower
      //  on the step than the compiler's own order, three rounds on one box)
    } else {
      SplitFrags<NP> f;
      split_tile_read<NP, 2 * SCH, 2 * SCH>(As, Bs, f);
      split_tile_mfma<NP>(f, acc);
      split_store(xsplit, Bst + (cur ^ 1) * SIMG);
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * NP, 0);   // all fragment reads first, in first-use order
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
    // weight tile t+1 landed (8 loads of this step are younger), own ds_writes done, the loads of t+2 in flight
    if (dmaA && ldB) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < T; t += 2) {
    step(t, 0, xb[0], xb[1]);
    if (t + 1 < T) step(t + 1, 1, xb[1], xb[0]);
  }
#undef USE_X
  if (live) {
    split_unflip(acc, wn);
    if constexpr (NP == 2) split_unscale(acc, inv_a, inv_b);
    gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
  }
}

// (A 256 x 256 workgroup tile for the six-product kernel - sixteen waves, TWO copies of the arrangement above sharing the
//  STAGED activation tiles, so that a 16 x 256 fp32 tile is fetched from L2 and split into its bf16 planes once per 256 output
//  rows: 40 instead of 56 KB through L2 per 256 x 256 x 16 and half the split arithmetic per MFMA; the copies take turns
//  fetching / splitting; 96 KB of LDS, one 1024-thread workgroup per CU; bit-identical results - was built in round 6
//  (`pw_gemm_split_quad_kernel`; the structure is all that is recorded) and measured on the training step: 156.2 / 157.0 against 154.8 / 155.0 ms,
//  same box.  Fewer L2 bytes and fewer VALU operations per MFMA buy nothing: what bounds these kernels is the matrix pipe
//  under the chip's power budget (busy x clock), as the yardstick of DESIGN.md 4.1 says.  Removed; profiles/r06_gemm_quad.txt.)
// wgrad: dW[M,N'] = sum over (sample, p) A[m][p] B[n][p], both operands p-contiguous fp32, both split
// in registers.  Thread t stages 8 consecutive p of row t>>1 (k-half t&1) of each operand.
// Needs K % 16 == 0 and 16-B aligned rows (host-checked; otherwise the f32 kernels run).
// Same pipeline as the fwd/dgrad kernel: loads two tiles ahead into alternating register sets (inline asm,
// counted waits), the two splits of tile t+1 between the MFMAs of tile t, raw barriers, one MFMA block
// per tile.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PARADIS_GEMM_BF16 forward / dgrad (the reference's bf16-mixed mode): operands rounded to bf16, ONE
// product.  The 128 x 256 workgroup tile of pw_gemm_split_wide_kernel (two 128-column halves sharing one weight tile),
// but with 32-deep k-tiles: with one product per k the k16 structure leaves four MFMAs per wave between two barriers and
// 32 KB of fp32 activations in flight per workgroup - the kernel then waits on its own per-tile chain, not on the
// matrix pipe or on bytes.  Here a tile is two k16 SLICES: eight MFMAs per wave and barrier, sixteen loads per thread and
// tile in flight two tiles ahead.  Images: weights [m-tile][k32-tile][slice][k-half][128 rows] chunks of 8 bf16
// (= two consecutive k16 tiles of the one-plane layout, K padded to a multiple of 32 with zeros), activations the same
// per stage in LDS.  48 KiB of LDS, <= 128 VGPRs: two 8-wave workgroups per CU.  Rows of the activation tile beyond K
// re-read row K - 1 against the zero padding of the weight image.
constexpr int BK32_SL = 2;                       // k16 slices per tile
template <bool C16 = false, bool ZM16 = false>       // bf16-stored output (and zout) / zmul: see gemm_epilogue
__global__ void __launch_bounds__(512, 4)
pw_gemm_bf16_k32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SIMG = simg(BK32_SL), SA = 2, KT = SBK * BK32_SL;
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [2 subs][2 activation stages][SIMG] | [SA weight stages][SIMG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = __builtin_amdgcn_readfirstlane(wave >> 2), lw = wave & 3, ltid = tid & 255;
  const int wm = lw >> 1, wn = lw & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN, NT2 = (NT + 1) / 2;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt2 = (L / MT) % NT2, bz = L / (MT * NT2);
  const bool live = 2 * nt2 + sub < NT;              // wave-uniform
  const int nt = min(2 * nt2 + sub, NT - 1);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = (g.K + KT - 1) / KT;

  // the weight tile (SIMG = 512 chunks of 16 bytes) goes by LDS-DMA, one chunk per thread
  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T * SIMG + tid;
  const int bh = __builtin_amdgcn_readfirstlane(ltid >> 7);      // k-half staged by this wave
  const float* Bb;
  {
    const uint64_t a = reinterpret_cast<uint64_t>(g.B + (int64_t)bz * g.b_bs);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    Bb = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
  }
  const int bn = min(n0 + (ltid & 127), g.N - 1);
  const uint32_t flip = split_flip_mask(ltid & 127);
  // ONE register set for the activations of the next tile (two sets of 2 x 8 next to 64 accumulators spilled 228 bytes
  // per lane at the 128 registers two workgroups per CU allow): tile t + 1 is loaded at the top of step t, converted
  // and stored behind the eight MFMAs of tile t - ~1,000 cycles at four waves per SIMD, the latency of an L2 hit
  float xb[BK32_SL][8] = {};
  auto issueA = [&](int t) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t * SIMG), (lds_ptr_t)(img + (4 + t % SA) * SIMG + wave * 64), 16, 0, 0);
  };
  auto round_store = [&](const float (&x)[BK32_SL][8], u32x4* o) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < BK32_SL; ++sl) {
      float xs[8];
      flip8(xs, x[sl], flip);       // sign checkerboard: odd 64-column blocks are staged negated
      o[sl * 2 * SCH] = round8(xs);
    }
  };
  const uint32_t boff = (uint32_t)bn * 4u;
  auto fetchB = [&](int t, float (&x)[BK32_SL][8]) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < BK32_SL; ++sl) {
      const int k0 = t * KT + sl * SBK + bh * 8;
      const float* p = Bb + (int64_t)min(k0, g.K - 1) * g.ldb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("global_load_dword %0, %1, %2" : "=&v"(x[sl][j]) : "v"(boff), "s"(p) : "memory");
        p += (k0 + j + 1 < g.K) ? g.ldb : 0;
      }
    }
  };
#define USE_X16(x, N) do { asm volatile("s_waitcnt vmcnt(" #N ")" :: "v"(x[0][0]), "v"(x[0][1]), "v"(x[0][2]), "v"(x[0][3]), \
                                        "v"(x[0][4]), "v"(x[0][5]), "v"(x[0][6]), "v"(x[0][7]), "v"(x[1][0]), "v"(x[1][1]),     \
                                        "v"(x[1][2]), "v"(x[1][3]), "v"(x[1][4]), "v"(x[1][5]), "v"(x[1][6]), "v"(x[1][7])      \
                                        : "memory");                                                                           \
                           __builtin_amdgcn_sched_barrier(0); } while (0)
  u32x4* const Bst = img + sub * 2 * SIMG + bh * SCH + (ltid & 127);   // this thread's chunk in its sub's stage 0

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issueA(0);
  fetchB(0, xb);
  USE_X16(xb, 0);
  round_store(xb, Bst);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    const u32x4* As = img + (4 + cur) * SIMG + lh * SCH + wm * 64 + li;
    const u32x4* Bs = img + (sub * 2 + cur) * SIMG + lh * SCH + wn * 64 + li;
    const bool more = t + 1 < T;                      // workgroup-uniform
    if (more) { issueA(t + 1); fetchB(t + 1, xb); }
#pragma unroll
    for (int sl = 0; sl < BK32_SL; ++sl) {          // one slice's fragments at a time: 16 registers
      const u32x4 a0 = As[sl * 2 * SCH], a1 = As[sl * 2 * SCH + 32], b0 = Bs[sl * 2 * SCH], b1 = Bs[sl * 2 * SCH + 32];
      SPLIT_MFMA(a0, b0, acc[0][0]); SPLIT_MFMA(a0, b1, acc[0][1]);
      SPLIT_MFMA(a1, b0, acc[1][0]); SPLIT_MFMA(a1, b1, acc[1][1]);
    }
    if (more) {
      USE_X16(xb, 0);                                 // (the DMA piece of this step is older than the loads: landed too)
      round_store(xb, Bst + (cur ^ 1) * SIMG);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
#undef USE_X16
  if (live) {
    split_unflip(acc, wn);
    gemm_epilogue<true, C16, ZM16>(g, acc, bz, m0, n0, wm, wn, li, lh);
  }
}

// The same kernel in 256 x 256 x 64 steps on sixteen waves (round 6; the structure of pw_gemm_b16_quad_kernel below, for an
// fp32-STORED activation operand): 1024 threads = two copies (msub 0 / 1 = m-tiles 2 mt2, 2 mt2 + 1) of the 8-wave arrangement
// sharing the ROUNDED activation tiles in LDS.  Per step each thread fetches the sixteen fp32 values it fetched per k32 tile
// above - copy msub stages the k-rows [32 msub, 32 msub + 32) of the 64 - so the rounding work and the activation bytes from
// L2 per MFMA halve, and a wave runs SIXTEEN MFMAs between two barriers.  Weight tiles: two k32 image tiles per copy and
// step by LDS-DMA.  Two stages of 64 KB.  An odd number of k32 image tiles: the last step runs two of its four slices.
constexpr int Q32_BCH = 4 * 2 * SCH;                      // chunks of one sub's activation image per stage: [4 slices][2 k-halves][128 columns]
constexpr int Q32_STAGE = 2 * (2 * simg(2)) + 2 * Q32_BCH;  // [copy 0: 2 k32 weight tiles | copy 1 | sub 0 | sub 1]
constexpr size_t q32_lds_bytes() { return (size_t)2 * Q32_STAGE * 16; }
template <bool C16, bool ZM16>
__global__ void __launch_bounds__(1024, 4)
pw_gemm_bf16_quad32_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int ACH = simg(2);                      // 512 chunks: one k32 weight-image tile
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int msub = wave >> 3, w8 = wave & 7, sub = w8 >> 2, lw = w8 & 3, ltid = tid & 255;
  const int wm = lw >> 1, wn = lw & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, MT2 = (MT + 1) / 2, NT = (g.N + BN - 1) / BN, NT2 = (NT + 1) / 2;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt2 = L % MT2, nt2 = (L / MT2) % NT2, bz = L / (MT2 * NT2);
  const bool live = 2 * nt2 + sub < NT && 2 * mt2 + msub < MT;       // wave-uniform
  const int nt = min(2 * nt2 + sub, NT - 1), mt = min(2 * mt2 + msub, MT - 1);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T32 = (g.K + 31) / 32, T = (T32 + 1) / 2;

  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T32 * ACH + (tid & 511);
  const int bh = __builtin_amdgcn_readfirstlane(ltid >> 7);      // k-half staged by this wave
  const float* Bb;
  {
    const uint64_t a = reinterpret_cast<uint64_t>(g.B + (int64_t)bz * g.b_bs);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    Bb = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
  }
  const int bn = min(n0 + (ltid & 127), g.N - 1);
  const uint32_t flip = split_flip_mask(ltid & 127);
  float xb[2][8] = {};
  auto issueA = [&](int t) __attribute__((always_inline)) {
    u32x4* st = img + (t & 1) * Q32_STAGE + 2 * msub * ACH;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t32 = min(2 * t + h, T32 - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t32 * ACH), (lds_ptr_t)(st + h * ACH + w8 * 64), 16, 0, 0);
    }
  };
  const uint32_t boff = (uint32_t)bn * 4u;
  auto fetchB = [&](int t, float (&x)[2][8]) __attribute__((always_inline)) {      // this copy's 32 k-rows of step t
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int k0 = t * 64 + msub * 32 + sl * SBK + bh * 8;
      const float* p = Bb + (int64_t)min(k0, g.K - 1) * g.ldb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("global_load_dword %0, %1, %2" : "=&v"(x[sl][j]) : "v"(boff), "s"(p) : "memory");
        p += (k0 + j + 1 < g.K) ? g.ldb : 0;
      }
    }
  };
#define USE_XQ16(x) do { asm volatile("s_waitcnt vmcnt(0)" :: "v"(x[0][0]), "v"(x[0][1]), "v"(x[0][2]), "v"(x[0][3]), \
                                      "v"(x[0][4]), "v"(x[0][5]), "v"(x[0][6]), "v"(x[0][7]), "v"(x[1][0]), "v"(x[1][1]),     \
                                      "v"(x[1][2]), "v"(x[1][3]), "v"(x[1][4]), "v"(x[1][5]), "v"(x[1][6]), "v"(x[1][7])      \
                                      : "memory");                                                                           \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
  // this thread's chunk in its sub's image of stage 0: slices 2 msub, 2 msub + 1
  u32x4* const Bst = img + 4 * ACH + sub * Q32_BCH + (2 * msub) * 2 * SCH + bh * SCH + (ltid & 127);
  auto round_store = [&](const float (&x)[2][8], u32x4* o) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      float xs[8];
      flip8(xs, x[sl], flip);
      o[sl * 2 * SCH] = round8(xs);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issueA(0);
  fetchB(0, xb);
  USE_XQ16(xb);
  round_store(xb, Bst);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    const u32x4* As = img + cur * Q32_STAGE + 2 * msub * ACH + lh * SCH + wm * 64 + li;
    const u32x4* Bs = img + cur * Q32_STAGE + 4 * ACH + sub * Q32_BCH + lh * SCH + wn * 64 + li;
    const bool more = t + 1 < T;                      // workgroup-uniform
    if (more) { issueA(t + 1); fetchB(t + 1, xb); }
    const int nsl = (2 * t + 1 < T32) ? 4 : 2;       // workgroup-uniform
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nsl) {
        const u32x4 a0 = As[sl * 2 * SCH], a1 = As[sl * 2 * SCH + 32], b0 = Bs[sl * 2 * SCH], b1 = Bs[sl * 2 * SCH + 32];
        SPLIT_MFMA(a0, b0, acc[0][0]); SPLIT_MFMA(a0, b1, acc[0][1]);
        SPLIT_MFMA(a1, b0, acc[1][0]); SPLIT_MFMA(a1, b1, acc[1][1]);
      }
    }
    if (more) {
      USE_XQ16(xb);                                   // (this step's DMAs are older than the loads: landed too)
      round_store(xb, Bst + (cur ^ 1) * Q32_STAGE);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
#undef USE_XQ16
  if (live) {
    split_unflip(acc, wn);
    gemm_epilogue<true, C16, ZM16>(g, acc, bz, m0, n0, wm, wn, li, lh);
  }
}

// PARADIS_GEMM_BF16 forward / dgrad with the activation operand STORED as bf16 (round 6; GemmArgs::io16 & IO_B16).
// B is [K rows][N columns] of bf16, n-contiguous (a [C, H W] plane stack as it sits in HBM).  Nothing of it passes through
// the vector ALU: a 32 x 256 tile (16 KB) goes HBM -> LDS by LDS-DMA, sixteen bytes per lane, and the MFMA's B fragment -
// eight consecutive k of one column per lane - comes out of the row-major image through the hardware transpose read
// ds_read_b64_tr_b16 (four k per read).  The weight tile is the [m-tile][k32-tile][slice][k-half][128 rows] image of
// pw_gemm_bf16_k32_kernel, by LDS-DMA as there.  Same 128 x 256 workgroup tile, accumulator layout and epilogue.
//   LDS: three stages of (8 KB weights + 16 KB activations) = 72 KB: two 8-wave workgroups per CU; tile t + 2 is in
//   flight while tile t is multiplied; one barrier per k-tile.
//   Image of the activation tile: row r (k) = 512 bytes, 16-byte chunk cc of the row stored at slot cc ^ ((r & 3) << 2):
//   the DMA writes lane-linearly (the permutation sits in the SOURCE address of a lane), and the four rows a transposed
//   read gathers per 16-lane group fall into the four bank quarters (conflict-free: rows 512 bytes apart would share one).
//   Requires N % 8 == 0, ldb % 8 == 0 and 16-byte aligned planes (host-checked).  Rows of a tile beyond K re-read row
//   K - 1 against the zero padding of the weight image; columns beyond N re-read the last eight and are never stored.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_t;
constexpr int B16_KT = 32, B16_ST = 3;
constexpr int B16_ACH = simg(2);                  // 512 chunks: the weight tile of one k32 step
constexpr int B16_BCH = B16_KT * 32;              // 1024 chunks: 32 k-rows x 256 columns of bf16
constexpr int B16_STAGE = B16_ACH + B16_BCH;      // chunks per stage (24 KB)
constexpr size_t b16_lds_bytes() { return (size_t)B16_ST * B16_STAGE * 16; }
template <bool C16, bool ZM16>
__global__ void __launch_bounds__(512, 4)
pw_gemm_b16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wave >> 2, lw = wave & 3;
  const int wm = lw >> 1, wn = lw & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN, NT2 = (NT + 1) / 2;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt2 = (L / MT) % NT2, bz = L / (MT * NT2);
  const bool live = 2 * nt2 + sub < NT;              // wave-uniform
  const int nt = min(2 * nt2 + sub, NT - 1);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T = (g.K + B16_KT - 1) / B16_KT;

  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T * B16_ACH + tid;
  const uint16_t* Bb = reinterpret_cast<const uint16_t*>(g.B) + (int64_t)bz * g.b_bs;
  // this lane's two source chunks of a tile: LDS chunk c = (2 wave + j) 64 + lane -> row c >> 5, slot c & 31
  int brow[2], bcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (2 * wave + j) * 64 + lane, r = c >> 5, slot = c & 31;
    brow[j] = r;
    bcol[j] = min(nt2 * 2 * BN + 8 * (slot ^ ((r & 3) << 2)), g.N - 8);
  }
  auto issue = [&](int t) __attribute__((always_inline)) {
    u32x4* st = img + (t % B16_ST) * B16_STAGE;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t * B16_ACH), (lds_ptr_t)(st + wave * 64), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = min(t * B16_KT + brow[j], g.K - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bb + (int64_t)k * g.ldb + bcol[j]),
                                       (lds_ptr_t)(st + B16_ACH + (2 * wave + j) * 64), 16, 0, 0);
    }
  };
  // transposed reads: lane 4q + p of 16-lane group gq supplies (row 8 lh + q [+ 16 slice + 4 e], columns 4p .. 4p + 3 of the
  // group's 16): byte offset of this lane inside a stage's activation image, one per 32-column block tn of the wave
  const int gq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  uint32_t boff[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int nbi = sub * 4 + wn * 2 + tn;                               // 32-column block inside the 256 columns
    boff[tn] = (uint32_t)((8 * lh + q4) * 512 + (4 * (nbi ^ q4) + 2 * (gq & 1) + (p4 >> 1)) * 16 + 8 * (p4 & 1));
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  if (T > 1) issue(1);
  for (int t = 0; t < T; ++t) {
    // tile t has landed (three DMAs per tile and lane; the next tile's may stay in flight) and every wave is past tile t - 1
    if (t + 1 < T) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 2 < T) issue(t + 2);
    const u32x4* st = img + (t % B16_ST) * B16_STAGE;
    const u32x4* As = st + lh * SCH + wm * 64 + li;
    const char* Bs = reinterpret_cast<const char*>(st + B16_ACH);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const u32x4 a0 = As[sl * 2 * SCH], a1 = As[sl * 2 * SCH + 32];
      u32x4 b[2];
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const char* pb = Bs + boff[tn] + sl * 16 * 512;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(pb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(pb + 4 * 512));
        const uint64_t l64 = __builtin_bit_cast(uint64_t, lo), h64 = __builtin_bit_cast(uint64_t, hi);
        b[tn] = (u32x4){(uint32_t)l64, (uint32_t)(l64 >> 32), (uint32_t)h64, (uint32_t)(h64 >> 32)};
      }
      SPLIT_MFMA(a0, b[0], acc[0][0]); SPLIT_MFMA(a0, b[1], acc[0][1]);
      SPLIT_MFMA(a1, b[0], acc[1][0]); SPLIT_MFMA(a1, b[1], acc[1][1]);
    }
  }
  if (live) {
    // the weight image holds the rows of odd 32-row blocks negated (sign checkerboard of the register-staged kernels; the
    // activations come straight from memory here, un-negated): block tm = 1 of every wave accumulated -C
    if (SPLIT_SIGNED) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][tn][r] = -acc[1][tn][r];
    }
    gemm_epilogue<true, C16, ZM16>(g, acc, bz, m0, n0, wm, wn, li, lh);
  }
}

// 256 x 256 x 64 steps with sixteen waves (round 6): 1024 threads = two copies (msub 0 / 1 = m-tiles 2 mt2, 2 mt2 + 1) of the
// 8-wave arrangement above sharing the activation tile in LDS, and k-steps of 64: SIXTEEN MFMAs per wave between two
// barriers instead of eight - the 128 x 256 x 32 kernel waits on its per-step chain (DMA landing, barrier, fragment reads;
// matrix pipe 19 % busy at 2.4 GHz), not on bytes.  Every wave keeps its 64 x 64 accumulators and its epilogue; sixteen waves
// per CU as with two 512-thread workgroups.  Two stages of 64 KB (32 KB of weight tiles - two k32 image tiles per copy - and
// 32 KB of activations); a thread issues four LDS-DMAs per step.  An odd number of k32 image tiles: the last step runs
// two of its four k16 slices.
constexpr int B16Q_KT = 64;
constexpr int B16Q_STAGE = 2 * (2 * B16_ACH) + 2 * B16_BCH;          // chunks per stage: [copy 0: 2 k32 weight tiles | copy 1 | 64 rows of activations]
constexpr size_t b16q_lds_bytes() { return (size_t)2 * B16Q_STAGE * 16; }
template <bool C16, bool ZM16>
__global__ void __launch_bounds__(1024, 4)
pw_gemm_b16_quad_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int msub = wave >> 3, w8 = wave & 7, sub = w8 >> 2, lw = w8 & 3;
  const int wm = lw >> 1, wn = lw & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, MT2 = (MT + 1) / 2, NT = (g.N + BN - 1) / BN, NT2 = (NT + 1) / 2;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt2 = L % MT2, nt2 = (L / MT2) % NT2, bz = L / (MT2 * NT2);
  const bool live = 2 * nt2 + sub < NT && 2 * mt2 + msub < MT;       // wave-uniform
  const int nt = min(2 * nt2 + sub, NT - 1), mt = min(2 * mt2 + msub, MT - 1);
  const int m0 = mt * BM, n0 = nt * BN;
  const int T32 = (g.K + B16_KT - 1) / B16_KT, T = (T32 + 1) / 2;     // k32 image tiles, k64 steps

  const u32x4* Ag = reinterpret_cast<const u32x4*>(g.A) + (int64_t)bz * g.a_bs + (int64_t)mt * T32 * B16_ACH + (tid & 511);
  const uint16_t* Bb = reinterpret_cast<const uint16_t*>(g.B) + (int64_t)bz * g.b_bs;
  // this lane's two source chunks of the activation tile: LDS chunk c = (2 wave + j) 64 + lane -> row c >> 5 (0..63), slot c & 31
  int brow[2], bcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (2 * wave + j) * 64 + lane, r = c >> 5, slot = c & 31;
    brow[j] = r;
    bcol[j] = min(nt2 * 2 * BN + 8 * (slot ^ ((r & 3) << 2)), g.N - 8);
  }
  auto issue = [&](int t) __attribute__((always_inline)) {
    u32x4* st = img + (t & 1) * B16Q_STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {        // the two k32 image tiles of the step (an odd T32: the last one twice, second use skipped)
      const int t32 = min(2 * t + h, T32 - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ag + (int64_t)t32 * B16_ACH),
                                       (lds_ptr_t)(st + (2 * msub + h) * B16_ACH + w8 * 64), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = min(t * B16Q_KT + brow[j], g.K - 1);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Bb + (int64_t)k * g.ldb + bcol[j]),
                                       (lds_ptr_t)(st + 4 * B16_ACH + (2 * wave + j) * 64), 16, 0, 0);
    }
  };
  const int gq = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  uint32_t boff[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int nbi = sub * 4 + wn * 2 + tn;
    boff[tn] = (uint32_t)((8 * lh + q4) * 512 + (4 * (nbi ^ q4) + 2 * (gq & 1) + (p4 >> 1)) * 16 + 8 * (p4 & 1));
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue(0);
  for (int t = 0; t < T; ++t) {
    // step t has landed (the only DMAs in flight) and every wave is past step t - 1: refill that stage at once
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < T) issue(t + 1);
    const u32x4* st = img + (t & 1) * B16Q_STAGE;
    const u32x4* As = st + 2 * msub * B16_ACH + lh * SCH + wm * 64 + li;
    const char* Bs = reinterpret_cast<const char*>(st + 4 * B16_ACH);
    const int nsl = (2 * t + 1 < T32) ? 4 : 2;       // workgroup-uniform
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nsl) {
        const u32x4 a0 = As[sl * 2 * SCH], a1 = As[sl * 2 * SCH + 32];     // (slice sl of the two k32 image tiles: 2 SCH chunks apart)
        u32x4 b[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const char* pb = Bs + boff[tn] + sl * 16 * 512;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(pb));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(pb + 4 * 512));
          const uint64_t l64 = __builtin_bit_cast(uint64_t, lo), h64 = __builtin_bit_cast(uint64_t, hi);
          b[tn] = (u32x4){(uint32_t)l64, (uint32_t)(l64 >> 32), (uint32_t)h64, (uint32_t)(h64 >> 32)};
        }
        SPLIT_MFMA(a0, b[0], acc[0][0]); SPLIT_MFMA(a0, b[1], acc[0][1]);
        SPLIT_MFMA(a1, b[0], acc[1][0]); SPLIT_MFMA(a1, b[1], acc[1][1]);
      }
    }
  }
  if (live) {
    if (SPLIT_SIGNED) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][tn][r] = -acc[1][tn][r];
    }
    gemm_epilogue<true, C16, ZM16>(g, acc, bz, m0, n0, wm, wn, li, lh);
  }
}

// (A 256 x 256 workgroup tile for this kernel - 2 x 4 waves of 128 x 64, 200-219 registers, one workgroup per CU, four
//  32 KB stages - halves the activation bytes pulled from L2 and was built and measured in round 6: 309 against 218 us at
//  896 x 896, the bf16-mixed step 83.8 against 77.6 ms.  At two waves per SIMD the store-bound epilogue doubles - a K = 32
//  launch 161 against 78 us - and takes back more than the k-loop gains; the weight gradient, whose epilogue writes one
//  fp32 tile per slab, does gain from that tile: pw_gemm_wgrad_square_kernel.  Removed; profiles/r06_amp_gemm_tiles.txt.
//  The same 256 x 256 tile with SIXTEEN waves - 1024 threads, two copies of the arrangement above sharing the activation tile
//  in LDS, every wave keeping its 64 x 64 accumulators and epilogue, 0.47 instead of 0.94 GB of activations per launch - was
//  parity-green and changed nothing: 183 / 210 against 180 / 222 us at 896^2 / 1024^2, the step 78.7 against 77.8 ms.  With
//  8 MFMAs per wave between two barriers the kernel waits on its per-tile chain (DMA landing, barrier, fragment reads), not on
//  L2 bytes - which is why the weight gradient, whose taller tiles also DOUBLE the MFMAs per barrier, gained and this did not.
//  With k-steps of 64 on top - sixteen MFMAs per barrier - it does pay: pw_gemm_b16_quad_kernel above.)
// (A soft rendezvous of a K-range slab's tiles - round 5: FETCH_SIZE 8.48 -> 4.06 GB per launch at 128 x 256, kernel 13 %
//  slower - was measured and removed: DESIGN_HISTORY.md section 4.1d, profiles/r05_wgrad_rendezvous.txt.)
template <int NP>
__global__ void __launch_bounds__(256, 3)
pw_gemm_wgrad_split_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SIMGP = simgp(NP);
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [2 stages][A|B][SIMGP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = g.K / SBK;
  const int64_t total = (int64_t)g.inner * KT;
  const int t_begin = (int)(total * bz / g.nbatch);
  const int T = (int)(total * (bz + 1) / g.nbatch) - t_begin;

  const int srow = tid >> 1, sh = tid & 1;
  const float* Ag = g.A + (int64_t)min(m0 + srow, g.M - 1) * g.lda + sh * 8;
  const float* Bg = g.B + (int64_t)min(n0 + srow, g.N - 1) * g.ldb + sh * 8;

  // (sample, k-tile) of the next tile to fetch, advanced incrementally
  int f_ib = t_begin / KT, f_kt = t_begin - f_ib * KT;
  struct Regs { f32x4 a0, a1, b0, b1; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Regs r0{zero4, zero4, zero4, zero4}, r1 = r0;   // defined values: the surplus split of a one-tile range reads r1
  auto fetch = [&](Regs& r) __attribute__((always_inline)) {
    const float* a = Ag + (int64_t)f_ib * g.a_is + (int64_t)f_kt * SBK;
    const float* b = Bg + (int64_t)f_ib * g.b_is + (int64_t)f_kt * SBK;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.a0) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.a1) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.b0) : "v"(b) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.b1) : "v"(b) : "memory");
    if (++f_kt == KT) { f_kt = 0; ++f_ib; }
  };
  // at most N younger vector-memory operations outstanding; registers as asm inputs (see USE_X)
#define USE_R(r, N) do { asm volatile("s_waitcnt vmcnt(" #N ")" :: "v"(r.a0), "v"(r.a1), "v"(r.b0), "v"(r.b1) : "memory"); \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
  const bool do_rowsum = g.rowsum != nullptr && nt == 0;
  float rs = 0.f;
  float sc_a = 1.f, sc_b = 1.f, inv_a = 1.f, inv_b = 1.f;
  if constexpr (NP == 2) {
    scale_from_amax(reduce_amax_partials(g.a_amax), sc_a, inv_a);
    scale_from_amax(reduce_amax_partials(g.b_amax), sc_b, inv_b);
  }
  // the offset of the MFMA's accumulator alignment (see "sign checkerboard") cancels between an element's slabs
  const uint32_t slab_flip = (SPLIT_SIGNED_WGRAD && (bz & 1)) ? 0x80000000u : 0u;      // workgroup-uniform
  if constexpr (NP == 2) sc_a = __uint_as_float(__float_as_uint(sc_a) ^ slab_flip);
  auto split_store = [&](const Regs& r, int st, bool keep) __attribute__((always_inline)) {
    const float xa[8] = {r.a0.x, r.a0.y, r.a0.z, r.a0.w, r.a1.x, r.a1.y, r.a1.z, r.a1.w};
    const float xb[8] = {r.b0.x, r.b0.y, r.b0.z, r.b0.w, r.b1.x, r.b1.y, r.b1.z, r.b1.w};
    // bias gradient: row sums of the staged dY values (a select, not a product: the surplus split of the
    // last tile works on stale registers that may hold NaNs)
    const float add = ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
    rs += keep ? add : 0.f;
    u32x4* o = img + st * 2 * SIMGP + sh * SCHP + srow;
    if constexpr (NP == 3) {
      u32x4 ha, ma, la, hb, mb, lb;
      float xs[8];
      flip8(xs, xa, slab_flip);    // sign alternation by slab: odd K-ranges accumulate -dW
      split8(xs, ha, ma, la);
      split8(xb, hb, mb, lb);
      o[0] = ha; o[2 * SCHP] = ma; o[4 * SCHP] = la;
      o += SIMGP;
      o[0] = hb; o[2 * SCHP] = mb; o[4 * SCHP] = lb;
    } else if constexpr (NP == 1) {
      float xs[8];
      flip8(xs, xa, slab_flip);
      o[0] = round8(xs);
      o[SIMGP] = round8(xb);
    } else {
      u32x4 ha, la, hb, lb;
      split8_f16(xa, sc_a, ha, la);
      split8_f16(xb, sc_b, hb, lb);
      o[0] = ha; o[2 * SCHP] = la;
      o += SIMGP;
      o[0] = hb; o[2 * SCHP] = lb;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (T > 0) {
    fetch(r0);
    if (T > 1) { fetch(r1); USE_R(r0, 4); } else { USE_R(r0, 0); }
    split_store(r0, 0, do_rowsum);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto step = [&](int t, int cur, Regs& rload, Regs& rsplit) __attribute__((always_inline)) {
    const u32x4* As = img + cur * 2 * SIMGP + lh * SCHP + wm * 64 + li;
    const u32x4* Bs = img + (cur * 2 + 1) * SIMGP + lh * SCHP + wn * 64 + li;
    SplitFrags<NP> f;
    split_tile_read<NP, 2 * SCHP, 2 * SCHP>(As, Bs, f);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < T) { fetch(rload); USE_R(rsplit, 4); }
    else USE_R(rsplit, 0);
    // one basic block for every tile; the last tile's split is surplus (stage nobody reads, keep = 0)
    split_tile_mfma<NP>(f, acc);
    split_store(rsplit, cur ^ 1, do_rowsum && t + 1 < T);
    if constexpr (NP > 1) {
#pragma unroll
    for (int i = 0; i < (NP == 3 ? 24 : 12); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU of the two splits
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < T; t += 2) {
    step(t, 0, r0, r1);
    if (t + 1 < T) step(t + 1, 1, r1, r0);
  }
#undef USE_R
  if (do_rowsum) {
    rs += __shfl_xor(rs, 1, 64);
    const int m = m0 + srow;
    if (sh == 0 && m < g.M) g.rowsum[(int64_t)bz * g.M + m] = rs;
  }
  if (slab_flip) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
  }
  if constexpr (NP == 2) split_unscale(acc, inv_a, inv_b);
  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// PARADIS_GEMM_BF16 weight gradient with bf16-STORED operands (round 6): pw_gemm_wgrad_split_kernel<1> with the staging
// of a bf16 operand reduced to one 16-byte load per thread and k-tile - the eight values ARE the LDS chunk (no rounding,
// no packing; the slab's sign alternation is an XOR on the packed sign bits) - while an fp32 operand is rounded in
// registers as before.  A16: dY is bf16 (GemmArgs::io16 & IO_A16), B16: X is bf16 (IO_B16).  Same tiles, slabs, row sums
// and epilogue; rows need 16-byte alignment in their own element size (host-checked).
template <bool A16, bool B16>
__global__ void __launch_bounds__(256, 3)
pw_gemm_wgrad_b16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SIMGP = simgp(1);
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [2 stages][A|B][SIMGP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int MT = (g.M + BM - 1) / BM, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = g.K / SBK;
  const int64_t total = (int64_t)g.inner * KT;
  const int t_begin = (int)(total * bz / g.nbatch);
  const int T = (int)(total * (bz + 1) / g.nbatch) - t_begin;

  const int srow = tid >> 1, sh = tid & 1;
  // byte addresses: element size 2 or 4 per operand
  constexpr int EA = A16 ? 2 : 4, EB = B16 ? 2 : 4;
  const char* Ag = reinterpret_cast<const char*>(g.A) + ((int64_t)min(m0 + srow, g.M - 1) * g.lda + sh * 8) * EA;
  const char* Bg = reinterpret_cast<const char*>(g.B) + ((int64_t)min(n0 + srow, g.N - 1) * g.ldb + sh * 8) * EB;

  int f_ib = t_begin / KT, f_kt = t_begin - f_ib * KT;
  struct Regs { f32x4 a0, a1, b0, b1; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Regs r0{zero4, zero4, zero4, zero4}, r1 = r0;
  auto fetch = [&](Regs& r) __attribute__((always_inline)) {
    const char* a = Ag + ((int64_t)f_ib * g.a_is + (int64_t)f_kt * SBK) * EA;
    const char* b = Bg + ((int64_t)f_ib * g.b_is + (int64_t)f_kt * SBK) * EB;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.a0) : "v"(a) : "memory");
    if constexpr (!A16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.a1) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.b0) : "v"(b) : "memory");
    if constexpr (!B16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.b1) : "v"(b) : "memory");
    if (++f_kt == KT) { f_kt = 0; ++f_ib; }
  };
  constexpr int NL = (A16 ? 1 : 2) + (B16 ? 1 : 2);      // loads per fetch
#define USE_RN(r, S) do { asm volatile("s_waitcnt vmcnt(" S ")" :: "v"(r.a0), "v"(r.a1), "v"(r.b0), "v"(r.b1) : "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)
  auto wait_keep_one = [&](Regs& r) __attribute__((always_inline)) {      // the younger fetch may stay in flight
    if constexpr (NL == 4) USE_RN(r, "4"); else if constexpr (NL == 3) USE_RN(r, "3"); else USE_RN(r, "2");
  };
  const bool do_rowsum = g.rowsum != nullptr && nt == 0;
  float rs = 0.f;
  const uint32_t slab_flip = (SPLIT_SIGNED_WGRAD && (bz & 1)) ? 0x80000000u : 0u;      // workgroup-uniform
  const uint32_t slab_flip16 = slab_flip | (slab_flip >> 16);
  auto split_store = [&](const Regs& r, int st, bool keep) __attribute__((always_inline)) {
    u32x4* o = img + st * 2 * SIMGP + sh * SCHP + srow;
    if constexpr (A16) {
      const u32x4 c = __builtin_bit_cast(u32x4, r.a0);
      if (do_rowsum) {       // (workgroup-uniform) bias gradient: row sums of the staged dY values
        float add = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) add += __uint_as_float(c[i] << 16) + __uint_as_float(c[i] & 0xffff0000u);
        rs += keep ? add : 0.f;
      }
      o[0] = (u32x4){c[0] ^ slab_flip16, c[1] ^ slab_flip16, c[2] ^ slab_flip16, c[3] ^ slab_flip16};
    } else {
      const float xa[8] = {r.a0.x, r.a0.y, r.a0.z, r.a0.w, r.a1.x, r.a1.y, r.a1.z, r.a1.w};
      const float add = ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
      rs += keep ? add : 0.f;
      float xs[8];
      flip8(xs, xa, slab_flip);
      o[0] = round8(xs);
    }
    if constexpr (B16) {
      o[SIMGP] = __builtin_bit_cast(u32x4, r.b0);
    } else {
      const float xb[8] = {r.b0.x, r.b0.y, r.b0.z, r.b0.w, r.b1.x, r.b1.y, r.b1.z, r.b1.w};
      o[SIMGP] = round8(xb);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (T > 0) {
    fetch(r0);
    if (T > 1) { fetch(r1); wait_keep_one(r0); } else { USE_RN(r0, "0"); }
    split_store(r0, 0, do_rowsum);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto step = [&](int t, int cur, Regs& rload, Regs& rsplit) __attribute__((always_inline)) {
    const u32x4* As = img + cur * 2 * SIMGP + lh * SCHP + wm * 64 + li;
    const u32x4* Bs = img + (cur * 2 + 1) * SIMGP + lh * SCHP + wn * 64 + li;
    SplitFrags<1> f;
    split_tile_read<1, 2 * SCHP, 2 * SCHP>(As, Bs, f);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < T) { fetch(rload); wait_keep_one(rsplit); }
    else USE_RN(rsplit, "0");
    split_tile_mfma<1>(f, acc);
    split_store(rsplit, cur ^ 1, do_rowsum && t + 1 < T);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < T; t += 2) {
    step(t, 0, r0, r1);
    if (t + 1 < T) step(t + 1, 1, r1, r0);
  }
#undef USE_RN
  if (do_rowsum) {
    rs += __shfl_xor(rs, 1, 64);
    const int m = m0 + srow;
    if (sh == 0 && m < g.M) g.rowsum[(int64_t)bz * g.M + m] = rs;
  }
  if (slab_flip) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
  }
  gemm_epilogue(g, acc, bz, m0, n0, wm, wn, li, lh);
}

// PARADIS_GEMM_BF16 weight gradient on a 256 x 128 output tile (round 6).  With ONE product per k the 128 x 128 kernels
// above are bound by the bytes their workgroups pull from L2, not by the matrix pipe: every 128-row block of dY is read
// once per 128-column block of X and vice versa (1024 x 1024 at N = 65,536, bf16 operands: 2.1 GB per launch = the 250 us
// they take at the ~10 TB/s the L2 delivers to the CUs).  A tile twice as tall halves the re-reads of X: 1.6 GB.
// 512 threads = 4 (M) x 2 (N) waves of 64 x 64; thread t stages chunk (row t >> 1, k-half t & 1) of dY and, waves 0-3
// only, of X; A16 / B16 = the operand is stored as bf16 (one 16-byte load is the LDS chunk) or as fp32 (two loads,
// rounded in registers); otherwise the pipeline of pw_gemm_wgrad_b16_kernel: loads two tiles ahead, one barrier per tile,
// K-range slabs with alternating sign, fused row sums.
constexpr int TALL_PA = 256 + 8, TALL_PB = 128 + 8, TALL_STAGE = 2 * TALL_PA + 2 * TALL_PB;   // chunks
constexpr size_t tall_lds_bytes() { return (size_t)2 * TALL_STAGE * 16; }
template <bool A16, bool B16>
__global__ void __launch_bounds__(512, 4)
pw_gemm_wgrad_tall_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [2 stages][A: 2 x TALL_PA | B: 2 x TALL_PB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool stB = wave < 4;                          // wave-uniform: these waves also stage X

  const int MT = (g.M + 255) / 256, NT = (g.N + BN - 1) / BN;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * 256, n0 = nt * BN;
  const int KT = g.K / SBK;
  const int64_t total = (int64_t)g.inner * KT;
  const int t_begin = (int)(total * bz / g.nbatch);
  const int T = (int)(total * (bz + 1) / g.nbatch) - t_begin;

  const int srow = tid >> 1, sh = tid & 1;
  constexpr int EA = A16 ? 2 : 4, EB = B16 ? 2 : 4;
  const char* Ag = reinterpret_cast<const char*>(g.A) + ((int64_t)min(m0 + srow, g.M - 1) * g.lda + sh * 8) * EA;
  const char* Bg = reinterpret_cast<const char*>(g.B) + ((int64_t)min(n0 + (srow & 127), g.N - 1) * g.ldb + sh * 8) * EB;

  int f_ib = t_begin / KT, f_kt = t_begin - f_ib * KT;
  struct Regs { f32x4 a0, a1, b0, b1; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Regs r0{zero4, zero4, zero4, zero4}, r1 = r0;
  auto fetch = [&](Regs& r) __attribute__((always_inline)) {
    const char* a = Ag + ((int64_t)f_ib * g.a_is + (int64_t)f_kt * SBK) * EA;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.a0) : "v"(a) : "memory");
    if constexpr (!A16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.a1) : "v"(a) : "memory");
    if (stB) {
      const char* b = Bg + ((int64_t)f_ib * g.b_is + (int64_t)f_kt * SBK) * EB;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.b0) : "v"(b) : "memory");
      if constexpr (!B16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.b1) : "v"(b) : "memory");
    }
    if (++f_kt == KT) { f_kt = 0; ++f_ib; }
  };
  constexpr int NLA = A16 ? 1 : 2, NLB = B16 ? 1 : 2;
#define USE_RN(r, S) do { asm volatile("s_waitcnt vmcnt(" S ")" :: "v"(r.a0), "v"(r.a1), "v"(r.b0), "v"(r.b1) : "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)
  // the registers of the older fetch are complete; the younger fetch (NLA or NLA + NLB loads of this wave) stays in flight
  auto wait_keep_one = [&](Regs& r) __attribute__((always_inline)) {
    if (stB) {
      if constexpr (NLA + NLB == 4) USE_RN(r, "4"); else if constexpr (NLA + NLB == 3) USE_RN(r, "3"); else USE_RN(r, "2");
    } else {
      if constexpr (NLA == 2) USE_RN(r, "2"); else USE_RN(r, "1");
    }
  };
  const bool do_rowsum = g.rowsum != nullptr && nt == 0;
  float rs = 0.f;
  const uint32_t slab_flip = (SPLIT_SIGNED_WGRAD && (bz & 1)) ? 0x80000000u : 0u;      // workgroup-uniform
  const uint32_t slab_flip16 = slab_flip | (slab_flip >> 16);
  auto split_store = [&](const Regs& r, int st, bool keep) __attribute__((always_inline)) {
    u32x4* o = img + st * TALL_STAGE + sh * TALL_PA + srow;
    if constexpr (A16) {
      const u32x4 c = __builtin_bit_cast(u32x4, r.a0);
      if (do_rowsum) {
        float add = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) add += __uint_as_float(c[i] << 16) + __uint_as_float(c[i] & 0xffff0000u);
        rs += keep ? add : 0.f;
      }
      o[0] = (u32x4){c[0] ^ slab_flip16, c[1] ^ slab_flip16, c[2] ^ slab_flip16, c[3] ^ slab_flip16};
    } else {
      const float xa[8] = {r.a0.x, r.a0.y, r.a0.z, r.a0.w, r.a1.x, r.a1.y, r.a1.z, r.a1.w};
      const float add = ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
      rs += keep ? add : 0.f;
      float xs[8];
      flip8(xs, xa, slab_flip);
      o[0] = round8(xs);
    }
    if (stB) {
      u32x4* ob = img + st * TALL_STAGE + 2 * TALL_PA + sh * TALL_PB + srow;      // (srow < 128 in these waves)
      if constexpr (B16) {
        ob[0] = __builtin_bit_cast(u32x4, r.b0);
      } else {
        const float xb[8] = {r.b0.x, r.b0.y, r.b0.z, r.b0.w, r.b1.x, r.b1.y, r.b1.z, r.b1.w};
        ob[0] = round8(xb);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (T > 0) {
    fetch(r0);
    if (T > 1) { fetch(r1); wait_keep_one(r0); } else { USE_RN(r0, "0"); }
    split_store(r0, 0, do_rowsum);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto step = [&](int t, int cur, Regs& rload, Regs& rsplit) __attribute__((always_inline)) {
    const u32x4* As = img + cur * TALL_STAGE + lh * TALL_PA + wm * 64 + li;
    const u32x4* Bs = img + cur * TALL_STAGE + 2 * TALL_PA + lh * TALL_PB + wn * 64 + li;
    SplitFrags<1> f;
    split_tile_read<1, 0, 0>(As, Bs, f);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < T) { fetch(rload); wait_keep_one(rsplit); }
    else USE_RN(rsplit, "0");
    split_tile_mfma<1>(f, acc);
    split_store(rsplit, cur ^ 1, do_rowsum && t + 1 < T);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int t = 0; t < T; t += 2) {
    step(t, 0, r0, r1);
    if (t + 1 < T) step(t + 1, 1, r1, r0);
  }
#undef USE_RN
  if (do_rowsum) {
    rs += __shfl_xor(rs, 1, 64);
    const int m = m0 + srow;
    if (sh == 0 && m < g.M) g.rowsum[(int64_t)bz * g.M + m] = rs;
  }
  if (slab_flip) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
  }
  // the wave's 64 rows as rows of the 128-row tile at m0 + 128 (wm >> 1)
  if (m0 + (wm >> 1) * 128 < g.M) gemm_epilogue(g, acc, bz, m0 + (wm >> 1) * 128, n0, wm & 1, wn, li, lh);
}

// (The bf16x3 - fp32-width - weight gradient on this 256 x 128 tile, three planes per operand image and 24 MFMAs per wave
//  and k-tile in the 128 registers that two 8-wave workgroups per CU leave, was tried in round 6: the step went from 150 to
//  308 ms - the twelve operand fragments next to 64 accumulators spill - and it failed the accuracy test; removed.  The
//  six-product kernels are bound by the matrix pipe's power budget, not by L2 bytes: DESIGN.md section 4.1.)
// ... and on a 256 x 256 tile (1.07 GB): 4 x 2 waves of 64 x 128, 128 accumulator registers per lane, ONE workgroup per CU
// (two waves per SIMD: enough for a kernel that waits on L2 bytes, not on the matrix pipe); every thread stages one chunk of
// each operand.  1024 x 1024, bf16 operands: 296 us (128 x 128) -> 250 (256 x 128) -> 221 (256 x 256).  PARADIS_WGRAD_SQUARE=0 /
// PARADIS_WGRAD_TALL=0 select the smaller tiles (A/B runs).
constexpr int SQ_P = 256 + 8, SQ_STAGE = 4 * SQ_P;   // chunks
constexpr size_t sq_lds_bytes() { return (size_t)4 * SQ_STAGE * 16; }
template <bool A16, bool B16>
__global__ void __launch_bounds__(512, 2)
pw_gemm_wgrad_square_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4* img = reinterpret_cast<u32x4*>(lds);        // [4 stages][A: 2 x SQ_P | B: 2 x SQ_P]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  constexpr bool stB = true;

  const int MT = (g.M + 255) / 256, NT = (g.N + 255) / 256;
  int L;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int mt = L % MT, nt = (L / MT) % NT, bz = L / (MT * NT);
  const int m0 = mt * 256, n0 = nt * 256;
  const int KT = g.K / SBK;
  const int64_t total = (int64_t)g.inner * KT;
  const int t_begin = (int)(total * bz / g.nbatch);
  const int T = (int)(total * (bz + 1) / g.nbatch) - t_begin;

  const int srow = tid >> 1, sh = tid & 1;
  constexpr int EA = A16 ? 2 : 4, EB = B16 ? 2 : 4;
  const char* Ag = reinterpret_cast<const char*>(g.A) + ((int64_t)min(m0 + srow, g.M - 1) * g.lda + sh * 8) * EA;
  const char* Bg = reinterpret_cast<const char*>(g.B) + ((int64_t)min(n0 + srow, g.N - 1) * g.ldb + sh * 8) * EB;

  int f_ib = t_begin / KT, f_kt = t_begin - f_ib * KT;
  struct Regs { f32x4 a0, a1, b0, b1; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Regs r0{zero4, zero4, zero4, zero4}, r1 = r0;
  auto fetch = [&](Regs& r) __attribute__((always_inline)) {
    const char* a = Ag + ((int64_t)f_ib * g.a_is + (int64_t)f_kt * SBK) * EA;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.a0) : "v"(a) : "memory");
    if constexpr (!A16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.a1) : "v"(a) : "memory");
    if (stB) {
      const char* b = Bg + ((int64_t)f_ib * g.b_is + (int64_t)f_kt * SBK) * EB;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r.b0) : "v"(b) : "memory");
      if constexpr (!B16) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(r.b1) : "v"(b) : "memory");
    }
    if (++f_kt == KT) { f_kt = 0; ++f_ib; }
  };
  constexpr int NLA = A16 ? 1 : 2, NLB = B16 ? 1 : 2;
#define USE_RN(r, S) do { asm volatile("s_waitcnt vmcnt(" S ")" :: "v"(r.a0), "v"(r.a1), "v"(r.b0), "v"(r.b1) : "memory"); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)
  // the registers of the older fetch are complete; the younger fetch (NLA or NLA + NLB loads of this wave) stays in flight
  auto wait_keep_one = [&](Regs& r) __attribute__((always_inline)) {
    if (stB) {
      if constexpr (NLA + NLB == 4) USE_RN(r, "4"); else if constexpr (NLA + NLB == 3) USE_RN(r, "3"); else USE_RN(r, "2");
    } else {
      if constexpr (NLA == 2) USE_RN(r, "2"); else USE_RN(r, "1");
    }
  };
  const bool do_rowsum = g.rowsum != nullptr && nt == 0;
  float rs = 0.f;
  const uint32_t slab_flip = (SPLIT_SIGNED_WGRAD && (bz & 1)) ? 0x80000000u : 0u;      // workgroup-uniform
  const uint32_t slab_flip16 = slab_flip | (slab_flip >> 16);
  auto split_store = [&](const Regs& r, int st, bool keep) __attribute__((always_inline)) {
    u32x4* o = img + st * SQ_STAGE + sh * SQ_P + srow;
    if constexpr (A16) {
      const u32x4 c = __builtin_bit_cast(u32x4, r.a0);
      if (do_rowsum) {
        float add = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) add += __uint_as_float(c[i] << 16) + __uint_as_float(c[i] & 0xffff0000u);
        rs += keep ? add : 0.f;
      }
      o[0] = (u32x4){c[0] ^ slab_flip16, c[1] ^ slab_flip16, c[2] ^ slab_flip16, c[3] ^ slab_flip16};
    } else {
      const float xa[8] = {r.a0.x, r.a0.y, r.a0.z, r.a0.w, r.a1.x, r.a1.y, r.a1.z, r.a1.w};
      const float add = ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
      rs += keep ? add : 0.f;
      float xs[8];
      flip8(xs, xa, slab_flip);
      o[0] = round8(xs);
    }
    if (stB) {
      u32x4* ob = img + st * SQ_STAGE + 2 * SQ_P + sh * SQ_P + srow;      // (srow < 128 in these waves)
      if constexpr (B16) {
        ob[0] = __builtin_bit_cast(u32x4, r.b0);
      } else {
        const float xb[8] = {r.b0.x, r.b0.y, r.b0.z, r.b0.w, r.b1.x, r.b1.y, r.b1.z, r.b1.w};
        ob[0] = round8(xb);
      }
    }
  };

  f32x16 acc[2][2], acc2[2][2];     // columns wn 128 + [0, 64) and + [64, 128)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

  {
    // Two k-tiles per barrier (sixteen MFMAs per wave between two barriers instead of eight): a ring of FOUR stages, tile t in
    // stage t & 3.  Entering a pair (t, t + 1) both tiles are staged and the loads of t + 2 / t + 3 sit in r0 / r1; inside the
    // pair tile t + 2 is stored behind the MFMAs of t and tile t + 3 behind those of t + 1 (their stages were last read a pair
    // ago: every wave is past that pair's barrier), each followed by the fetch of the tile four ahead.
    auto stage_tile = [&](int t, Regs& r, bool younger_in_flight) __attribute__((always_inline)) {
      if (younger_in_flight) wait_keep_one(r); else USE_RN(r, "0");
      split_store(r, t & 3, do_rowsum);
    };
    if (T > 0) { fetch(r0); }
    if (T > 1) { fetch(r1); }
    if (T > 0) { stage_tile(0, r0, T > 1); asm volatile("" : "+v"(rs)); if (T > 2) fetch(r0); }      // (the pin: see half())
    if (T > 1) { stage_tile(1, r1, T > 2); asm volatile("" : "+v"(rs)); if (T > 3) fetch(r1); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    auto half = [&](int t, Regs& r) __attribute__((always_inline)) {      // tile t is staged; r holds the loads of tile t + 2
      const u32x4* As = img + (t & 3) * SQ_STAGE + lh * SQ_P + wm * 64 + li;
      const u32x4* Bs = img + (t & 3) * SQ_STAGE + 2 * SQ_P + lh * SQ_P + wn * 128 + li;
      SplitFrags<1> f, f2;
      split_tile_read<1, 0, 0>(As, Bs, f);
      f2.a[0][0] = f.a[0][0]; f2.a[0][1] = f.a[0][1];
      f2.b[0][0] = Bs[64]; f2.b[0][1] = Bs[96];
      __builtin_amdgcn_sched_barrier(0);
      split_tile_mfma<1>(f, acc);
      split_tile_mfma<1>(f2, acc2);
      if (t + 2 < T) {
        stage_tile(t + 2, r, t + 3 < T);       // (the fetch of tile t + 3, issued after this one's, may stay in flight)
        // The row sum must be COMPLETE before the next fetch is issued.  Left alone, the compiler sinks the add chain of an
        // fp32 dY below the (volatile, but register-only) load statements: the old value of r then lives across them, the new
        // loads get other registers and a copy "new -> old registers" follows the load at once - it reads registers whose data
        // has not arrived, and the data lands later in registers that hold addresses by then (NaNs, memory faults on long
        // slabs: this schedule's first version; tools/async_load_check.py finds such copies in the ISA).
        asm volatile("" : "+v"(rs));
        if (t + 4 < T) fetch(r);
      }
    };
    for (int t = 0; t < T; t += 2) {
      half(t, r0);
      if (t + 1 < T) half(t + 1, r1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
#undef USE_RN
  if (do_rowsum) {
    rs += __shfl_xor(rs, 1, 64);
    const int m = m0 + srow;
    if (sh == 0 && m < g.M) g.rowsum[(int64_t)bz * g.M + m] = rs;
  }
  if (slab_flip) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = -acc[i][j][r]; acc2[i][j][r] = -acc2[i][j][r]; }
  }
  // the wave's 64 rows as rows of the 128-row tile at m0 + 128 (wm >> 1)
  if (m0 + (wm >> 1) * 128 < g.M) {
    const int mm = m0 + (wm >> 1) * 128, nn = n0 + wn * 128;           // 64-column halves as "wn" 0 / 1 of a 128-column tile
    if (nn < g.N) {
      gemm_epilogue(g, acc, bz, mm, nn, wm & 1, 0, li, lh);
      gemm_epilogue(g, acc2, bz, mm, nn, wm & 1, 1, li, lh);
    }
  }
}

// out[i] = slabs[0][i] + slabs[1][i] + ... in that order; vec: n % 4 == 0 and 16-byte aligned pointers (four
// elements per thread, four slabs' loads in flight)
#if GEMM_PART < 2
__global__ void __launch_bounds__(256)
slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n, int S, int vec,
                   const float* __restrict__ slabs2, float* __restrict__ out2, int n2, int blocks1) {
  // second, small reduction riding in the same launch (the bias gradient's row sums next to the weight gradient's
  // slabs: 131 launches per training step less): blocks [blocks1, gridDim.x)
  if ((int)blockIdx.x >= blocks1) {
    const int i = ((int)blockIdx.x - blocks1) * 256 + threadIdx.x;
    if (i < n2) {
      float t = 0.f;
      for (int k = 0; k < S; ++k) t += slabs2[(int64_t)k * n2 + i];
      out2[i] = t;
    }
    return;
  }
  const int nblk = blocks1;      // (the grid-stride loops below run over the first `blocks1` blocks)
  if (vec) {
    const int64_t n4 = n >> 2;
    const float4* sl = reinterpret_cast<const float4*>(slabs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)nblk * 256) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = 0;
      for (; k + 4 <= S; k += 4) {
        float4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = sl[(int64_t)(k + j) * n4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s.x += q[j].x; s.y += q[j].y; s.z += q[j].z; s.w += q[j].w; }
      }
      for (; k < S; ++k) {
        const float4 q = sl[(int64_t)k * n4 + i];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
      reinterpret_cast<float4*>(out)[i] = s;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += slabs[(int64_t)k * n + i];
    out[i] = s;
  }
}
#endif

#if GEMM_PART < 2
int slots() { return 256 * g_wg_per_cu; }

// both operands p-contiguous with whole, 16-B aligned 16-float chunks (LDS-DMA and split kernels)
bool wgrad_vec_layout(int N, int64_t dy_bs, int64_t x_bs, const void* a, const void* b) {
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return N % DBK == 0 && (dy_bs & 3) == 0 && (x_bs & 3) == 0 && a16(a) && a16(b);
}
bool wgrad_dma_ok(int N, int64_t dy_bs, int64_t x_bs, const void* a, const void* b) {
  return g_wgrad_dma_stages >= 2 && wgrad_vec_layout(N, dy_bs, x_bs, a, b);
}

// number of k-range splits: one round of resident workgroups over the CUs
int wgrad_splits(int B, int M, int K, int N, int bk, int wg_per_cu) {
  const int tiles = ((M + BM - 1) / BM) * ((K + BN - 1) / BN);
  const int64_t total_kt = (int64_t)B * ((N + bk - 1) / bk);
  int s = (int)std::max<int64_t>(1, std::min<int64_t>(256 * wg_per_cu / tiles, total_kt));
  // the split kernels accumulate alternate slabs with opposite sign so that the bf16 MFMA's alignment offset cancels
  // in the slab sum ("sign checkerboard"): that takes an even number of slabs (1536 x 384: 21 -> 20; round 5)
  if (s > 1) s &= ~1;
  return s;
}
int wgrad_dma_wgs() { return g_wgrad_dma_stages == 2 ? 4 : 3; }
// bf16-mixed scheme: the 256 x 128 kernel where the taller tile wastes at most ~1/7 of its rows (PARADIS_WGRAD_TALL=0: off)
bool wgrad_tall_ok(int M) {
  static const bool on = [] { const char* e = getenv("PARADIS_WGRAD_TALL"); return !(e && e[0] == '0'); }();
  return on && M >= 256 && ((M + 255) / 256) * 256 * 7 <= M * 8;
}
bool wgrad_square_on() {
  static const bool on = [] { const char* e = getenv("PARADIS_WGRAD_SQUARE"); return !(e && e[0] == '0'); }();
  return on;
}
int wgrad_splits_square(int B, int M, int K, int N) {
  const int tiles = ((M + 255) / 256) * ((K + 255) / 256);
  const int64_t total_kt = (int64_t)B * ((N + SBK - 1) / SBK);
  int s = (int)std::max<int64_t>(1, std::min<int64_t>(256 / tiles, total_kt));
  if (s > 1) s &= ~1;
  return s;
}
int wgrad_splits_tall(int B, int M, int K, int N) {
  const int tiles = ((M + 255) / 256) * ((K + BN - 1) / BN);
  const int64_t total_kt = (int64_t)B * ((N + SBK - 1) / SBK);
  int s = (int)std::max<int64_t>(1, std::min<int64_t>(512 / tiles, total_kt));
  if (s > 1) s &= ~1;
  return s;
}

template <bool A_KC, bool B_KC, int BK>
int launch_gemm_bk(const GemmArgs& g, int grid, hipStream_t st) {
  // the dynamic-LDS request doubles as the occupancy control: 160 KiB / request = workgroups per CU
  size_t request = std::max(lds_bytes(BK), (size_t)(160 * 1024 / g_wg_per_cu) & ~(size_t)255);
  request = std::min(request, (size_t)160 * 1024);
  static PerDeviceOnce once;
  if (once.first()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_kernel<A_KC, B_KC, BK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess) {
      paradis_set_error("pw_gemm: cannot reserve LDS");
      return 2;
    }
  }
  hipLaunchKernelGGL((pw_gemm_kernel<A_KC, B_KC, BK>), dim3(grid), dim3(256), request, st, g);
  return 0;
}

bool dma_eligible(const GemmArgs& g) {
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g_dma_stages >= 2 && g.inner == 0 && g.K % DBK == 0 && g.M % 4 == 0 && g.N % 4 == 0 &&
         g.M >= 4 && g.N >= 4 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && (g.a_bs & 3) == 0 &&
         (g.b_bs & 3) == 0 && a16(g.A) && a16(g.B);
}

int launch_gemm_dma(const GemmArgs& g, int grid, hipStream_t st) {
  const size_t bytes = (size_t)g_dma_stages * 2 * DTILE * sizeof(float);
  switch (g_dma_stages) {
    case 2: hipLaunchKernelGGL((pw_gemm_dma_kernel<2, 4>), dim3(grid), dim3(256), bytes, st, g); break;
    case 3: hipLaunchKernelGGL((pw_gemm_dma_kernel<3, 3>), dim3(grid), dim3(256), bytes, st, g); break;
    default: hipLaunchKernelGGL((pw_gemm_dma_kernel<4, 2>), dim3(grid), dim3(256), (size_t)4 * 2 * DTILE * sizeof(float), st, g); break;
  }
  return 0;
}

template <bool A_KC, bool B_KC>
int launch_gemm(const GemmArgs& g, int grid, hipStream_t st) {
  return g_bk == 32 ? launch_gemm_bk<A_KC, B_KC, 32>(g, grid, st)
                    : launch_gemm_bk<A_KC, B_KC, 16>(g, grid, st);
}

constexpr size_t split_lds(int np) { return (size_t)(2 + split_astages(np)) * simg(np) * 16; }
constexpr size_t split_lds_wgrad(int np) { return (size_t)2 * 2 * simgp(np) * 16; }
constexpr int AMAX_WORDS = PARADIS_AMAX_PARTIALS;
// f16x2 weight image: the planes, then 16 bytes ([0] = bits of max |W|), then the amax partials of W
constexpr size_t F16_TAIL_BYTES = 16 + (size_t)AMAX_WORDS * 4;

// k16 tiles of an image: the one-plane (bf16) layout is read in pairs of tiles (pw_gemm_bf16_k32_kernel): an even count
int split_image_ktiles(int K, int np) {
  const int kt = (K + SBK - 1) / SBK;
  return np == 1 ? (kt + 1) & ~1 : kt;
}
int64_t split_image_chunks(int M, int K, int np = 3) {
  return (int64_t)((M + BM - 1) / BM) * split_image_ktiles(K, np) * simg(np);
}

bool known_scheme(int scheme) {
  return scheme == PARADIS_GEMM_EXACT || scheme == PARADIS_GEMM_BF16X3 || scheme == PARADIS_GEMM_F16X2 ||
         scheme == PARADIS_GEMM_BF16;
}

int launch_amax(const float* x, int B, int64_t inner, int64_t bs, uint32_t* out, hipStream_t st) {
  const int vec = (inner % 4 == 0) && (bs % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipLaunchKernelGGL(amax_partials_kernel, dim3(AMAX_WORDS), dim3(256), 0, st, x, B, inner, bs, vec, out);
  return 0;
}

template <int NP>
int launch_split_np(const GemmArgs& d, hipStream_t st) {
  const int grid = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * d.nbatch;
  static PerDeviceOnce once;
  if (split_lds(NP) > 64 * 1024 && once.first()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_split_kernel<NP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds(NP)) != hipSuccess) {
      paradis_set_error("pw_gemm(split): cannot reserve LDS");
      return 2;
    }
  }
  hipLaunchKernelGGL(pw_gemm_split_kernel<NP>, dim3(grid), dim3(256), split_lds(NP), st, d);
  return 0;
}
constexpr size_t split_wide_lds(int nsub, int np = 2) { return (size_t)(2 * nsub + 2) * simg(np) * 16; }   // f16x2: 48 KiB, bf16x3: 72 KiB
// n-tiles per workgroup: 2 (two 8-wave workgroups per CU).  4 - one 16-wave workgroup per CU, another 17 % fewer
// bytes - measured 2.5 % SLOWER: a single workgroup's waves all stop at the same barriers.
constexpr int SPLIT_WIDE_NSUB = 2;
// scheme: PARADIS_GEMM_BF16X3 or PARADIS_GEMM_F16X2 (the latter with d.a_amax / d.b_amax set)
// bf16x3 on the 128 x 256 tile (round 3).  With all twelve fragments of a k-tile live (the 128 x 128 kernel's way) the
// kernel needs ~150 registers and hipcc spills 1.2 KB per lane at the 128 that two 8-wave workgroups per CU allow;
// reading the B fragments one PLANE at a time (8 instead of 24 registers, planes in the order l, m, h) brings it to
// 128 registers and 12 bytes of scratch.  Training step 160.0 -> 156.3 ms, GEMMs 188 -> 194 TF (same box, two rounds).
#ifndef SPLIT_WIDE_BF16X3      // (0: the 128 x 128 kernel for every shape; A/B builds)
#define SPLIT_WIDE_BF16X3 1
#endif
template <int NP>
int launch_split_wide(const GemmArgs& d, int NT, hipStream_t st) {
  constexpr int NSUB = SPLIT_WIDE_NSUB;
  static PerDeviceOnce once;
  if (split_wide_lds(NSUB, NP) > 64 * 1024 && once.first()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm_split_wide_kernel<NSUB, NP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_wide_lds(NSUB, NP)) != hipSuccess) {
      paradis_set_error("pw_gemm(split): cannot reserve LDS");
      return 2;
    }
  }
  const int grid = ((d.M + BM - 1) / BM) * ((NT + NSUB - 1) / NSUB) * d.nbatch;     // 128 x (128 NSUB) tiles
  hipLaunchKernelGGL((pw_gemm_split_wide_kernel<NSUB, NP>), dim3(grid), dim3(256 * NSUB), split_wide_lds(NSUB, NP), st, d);
  return 0;
}
int launch_split(const GemmArgs& d, int scheme, hipStream_t st) {
  const int NT = (d.N + BN - 1) / BN;
  if (scheme == PARADIS_GEMM_BF16) return pd_amp_launch_fwd(d, st);      // (the other translation unit)
  if (scheme != PARADIS_GEMM_F16X2) {
#if SPLIT_WIDE_BF16X3
    if (NT >= 2) return launch_split_wide<3>(d, NT, st);
#endif
    return launch_split_np<3>(d, st);
  }
  if (NT < 2) return launch_split_np<2>(d, st);
  return launch_split_wide<2>(d, NT, st);
}

int check_gemm(const char* name, int B, int M, int K, int N) {
  PD_REQUIRE(B >= 0 && M >= 1 && K >= 1 && N >= 1, "%s: bad shape B=%d M=%d K=%d N=%d", name, B, M, K, N);
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * std::max(B, 1);
  PD_REQUIRE(tiles < (1ll << 31), "%s: too many tiles", name);
  return 0;
}

#endif   // GEMM_PART < 2
}  // namespace

// ---- launchers of the bf16-mixed kernels (called from the first translation unit) ---------------------------------------
#if GEMM_PART == 0 || GEMM_PART == 2
int pd_amp_launch_fwd(const GemmArgs& d, hipStream_t st) {
  const int NT = (d.N + BN - 1) / BN;
  // (round 5: the k16 kernels with one plane ran the bf16-mixed step at 91.6 ms; a 256 x 128 tile - two M-tiles
  //  sharing one fp32 activation tile, 16 instead of 20 KB through L2 per tile pair - at 95.9 ms: with four MFMAs per
  //  wave and barrier the kernel is bound by its per-tile latency chain, not by bytes.  Hence 32-deep tiles.)
  const size_t lds = (size_t)(2 * 2 + 2) * simg(BK32_SL) * 16;
  const int grid = ((d.M + BM - 1) / BM) * ((NT + 1) / 2) * d.nbatch;
  const bool c16 = (d.io16 & IO_C16) != 0, zm16 = (d.io16 & IO_ZM16) != 0 && d.zmul != nullptr;
  if (d.io16 & IO_B16) {     // activations stored as bf16: LDS-DMA + transposed reads (layout checked by the caller)
    static PerDeviceOnce once;
    if (once.first()) {
      const void* ks[4] = {reinterpret_cast<const void*>(&pw_gemm_b16_kernel<false, false>),
                           reinterpret_cast<const void*>(&pw_gemm_b16_kernel<true, false>),
                           reinterpret_cast<const void*>(&pw_gemm_b16_kernel<false, true>),
                           reinterpret_cast<const void*>(&pw_gemm_b16_kernel<true, true>)};
      for (const void* k : ks)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b16_lds_bytes()) != hipSuccess) {
          paradis_set_error("pw_gemm(b16): cannot reserve LDS");
          return 2;
        }
    }
    const int MT = (d.M + BM - 1) / BM;
    static const bool quad_on = [] { const char* e = getenv("PARADIS_GEMM_B16_QUAD"); return !(e && e[0] == '0'); }();   // (=0: A/B)
    if (quad_on && MT >= 2 && ((MT + 1) / 2) * 2 * 7 <= MT * 8) {       // (an odd MT repeats its last m-tile: at most 1/8)
      static PerDeviceOnce once_q;
      if (once_q.first()) {
        const void* ks[4] = {reinterpret_cast<const void*>(&pw_gemm_b16_quad_kernel<false, false>),
                             reinterpret_cast<const void*>(&pw_gemm_b16_quad_kernel<true, false>),
                             reinterpret_cast<const void*>(&pw_gemm_b16_quad_kernel<false, true>),
                             reinterpret_cast<const void*>(&pw_gemm_b16_quad_kernel<true, true>)};
        for (const void* k : ks)
          if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b16q_lds_bytes()) != hipSuccess) {
            paradis_set_error("pw_gemm(b16 quad): cannot reserve LDS");
            return 2;
          }
      }
      const dim3 gq(((MT + 1) / 2) * ((NT + 1) / 2) * d.nbatch), bq(1024);
      if (c16 && zm16) hipLaunchKernelGGL((pw_gemm_b16_quad_kernel<true, true>), gq, bq, b16q_lds_bytes(), st, d);
      else if (c16) hipLaunchKernelGGL((pw_gemm_b16_quad_kernel<true, false>), gq, bq, b16q_lds_bytes(), st, d);
      else if (zm16) hipLaunchKernelGGL((pw_gemm_b16_quad_kernel<false, true>), gq, bq, b16q_lds_bytes(), st, d);
      else hipLaunchKernelGGL((pw_gemm_b16_quad_kernel<false, false>), gq, bq, b16q_lds_bytes(), st, d);
      return 0;
    }
    const dim3 gr(grid), bl(512);
    if (c16 && zm16) hipLaunchKernelGGL((pw_gemm_b16_kernel<true, true>), gr, bl, b16_lds_bytes(), st, d);
    else if (c16) hipLaunchKernelGGL((pw_gemm_b16_kernel<true, false>), gr, bl, b16_lds_bytes(), st, d);
    else if (zm16) hipLaunchKernelGGL((pw_gemm_b16_kernel<false, true>), gr, bl, b16_lds_bytes(), st, d);
    else hipLaunchKernelGGL((pw_gemm_b16_kernel<false, false>), gr, bl, b16_lds_bytes(), st, d);
    return 0;
  }
  {
    const int MT = (d.M + BM - 1) / BM;
    static const bool q32_on = [] { const char* e = getenv("PARADIS_GEMM_K32_QUAD"); return !(e && e[0] == '0'); }();   // (=0: A/B)
    if (q32_on && MT >= 2 && ((MT + 1) / 2) * 2 * 7 <= MT * 8) {
      static PerDeviceOnce once_q32;
      if (once_q32.first()) {
        const void* ks[4] = {reinterpret_cast<const void*>(&pw_gemm_bf16_quad32_kernel<false, false>),
                             reinterpret_cast<const void*>(&pw_gemm_bf16_quad32_kernel<true, false>),
                             reinterpret_cast<const void*>(&pw_gemm_bf16_quad32_kernel<false, true>),
                             reinterpret_cast<const void*>(&pw_gemm_bf16_quad32_kernel<true, true>)};
        for (const void* k : ks)
          if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q32_lds_bytes()) != hipSuccess) {
            paradis_set_error("pw_gemm(k32 quad): cannot reserve LDS");
            return 2;
          }
      }
      const dim3 gq(((MT + 1) / 2) * ((NT + 1) / 2) * d.nbatch), bq(1024);
      if (c16 && zm16) hipLaunchKernelGGL((pw_gemm_bf16_quad32_kernel<true, true>), gq, bq, q32_lds_bytes(), st, d);
      else if (c16) hipLaunchKernelGGL((pw_gemm_bf16_quad32_kernel<true, false>), gq, bq, q32_lds_bytes(), st, d);
      else if (zm16) hipLaunchKernelGGL((pw_gemm_bf16_quad32_kernel<false, true>), gq, bq, q32_lds_bytes(), st, d);
      else hipLaunchKernelGGL((pw_gemm_bf16_quad32_kernel<false, false>), gq, bq, q32_lds_bytes(), st, d);
      return 0;
    }
  }
  if (c16 && zm16) hipLaunchKernelGGL((pw_gemm_bf16_k32_kernel<true, true>), dim3(grid), dim3(512), lds, st, d);
  else if (c16) hipLaunchKernelGGL((pw_gemm_bf16_k32_kernel<true, false>), dim3(grid), dim3(512), lds, st, d);
  else if (zm16) hipLaunchKernelGGL((pw_gemm_bf16_k32_kernel<false, true>), dim3(grid), dim3(512), lds, st, d);
  else hipLaunchKernelGGL((pw_gemm_bf16_k32_kernel<false, false>), dim3(grid), dim3(512), lds, st, d);
  return 0;
}
#endif

#if GEMM_PART == 0 || GEMM_PART == 3
int pd_amp_launch_wgrad(const GemmArgs& g0, int io16, int kind, int grid, hipStream_t st) {
  GemmArgs g = g0;
  g.io16 = io16;
  constexpr size_t lds128 = (size_t)2 * 2 * simgp(1) * 16;
  if (kind == 3) {
    const dim3 gr(grid), bl(512);
    const size_t ld = sq_lds_bytes();
    static PerDeviceOnce once_sq;
    if (once_sq.first()) {
      const void* ks[4] = {reinterpret_cast<const void*>(&pw_gemm_wgrad_square_kernel<false, false>),
                           reinterpret_cast<const void*>(&pw_gemm_wgrad_square_kernel<true, false>),
                           reinterpret_cast<const void*>(&pw_gemm_wgrad_square_kernel<false, true>),
                           reinterpret_cast<const void*>(&pw_gemm_wgrad_square_kernel<true, true>)};
      for (const void* k : ks)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ld) != hipSuccess) {
          paradis_set_error("pw_gemm_wgrad(square): cannot reserve LDS");
          return 2;
        }
    }
    if ((io16 & IO_A16) && (io16 & IO_B16)) hipLaunchKernelGGL((pw_gemm_wgrad_square_kernel<true, true>), gr, bl, ld, st, g);
    else if (io16 & IO_A16) hipLaunchKernelGGL((pw_gemm_wgrad_square_kernel<true, false>), gr, bl, ld, st, g);
    else if (io16 & IO_B16) hipLaunchKernelGGL((pw_gemm_wgrad_square_kernel<false, true>), gr, bl, ld, st, g);
    else hipLaunchKernelGGL((pw_gemm_wgrad_square_kernel<false, false>), gr, bl, ld, st, g);
  } else if (kind == 2) {
    const dim3 gr(grid), bl(512);
    const size_t ld = tall_lds_bytes();
    if ((io16 & IO_A16) && (io16 & IO_B16)) hipLaunchKernelGGL((pw_gemm_wgrad_tall_kernel<true, true>), gr, bl, ld, st, g);
    else if (io16 & IO_A16) hipLaunchKernelGGL((pw_gemm_wgrad_tall_kernel<true, false>), gr, bl, ld, st, g);
    else if (io16 & IO_B16) hipLaunchKernelGGL((pw_gemm_wgrad_tall_kernel<false, true>), gr, bl, ld, st, g);
    else hipLaunchKernelGGL((pw_gemm_wgrad_tall_kernel<false, false>), gr, bl, ld, st, g);
  } else if (kind == 1) {
    if ((io16 & IO_A16) && (io16 & IO_B16))
      hipLaunchKernelGGL((pw_gemm_wgrad_b16_kernel<true, true>), dim3(grid), dim3(256), lds128, st, g);
    else if (io16 & IO_A16)
      hipLaunchKernelGGL((pw_gemm_wgrad_b16_kernel<true, false>), dim3(grid), dim3(256), lds128, st, g);
    else
      hipLaunchKernelGGL((pw_gemm_wgrad_b16_kernel<false, true>), dim3(grid), dim3(256), lds128, st, g);
  } else {
    hipLaunchKernelGGL(pw_gemm_wgrad_split_kernel<1>, dim3(grid), dim3(256), lds128, st, g);
  }
  return 0;
}
#endif

#if GEMM_PART < 2
#ifdef PARADIS_DEV_KNOBS
// diagnostic knobs of the development build only (`make dev`, tools/gemm_bench.py); the shipped
// library exports none of them.  bk in {16,32}, wg_per_cu in 1..4
extern "C" void paradis_debug_set_gemm(int bk, int wg_per_cu) {
  if (bk == 16 || bk == 32) g_bk = bk;
  if (wg_per_cu >= 1 && wg_per_cu <= 4) g_wg_per_cu = wg_per_cu;
}
extern "C" void paradis_debug_set_gemm_stagger(int units) { g_stagger = units < 0 ? 0 : units; }
extern "C" void paradis_debug_set_gemm_dma(int stages) { g_dma_stages = stages < 2 ? 0 : (stages > 4 ? 4 : stages); }
extern "C" void paradis_debug_set_wgrad_dma(int stages) { g_wgrad_dma_stages = stages < 2 ? 0 : (stages > 3 ? 3 : stages); }
#endif

extern "C" size_t paradis_pw_gemm_split_bytes(int M, int K, int scheme) {
  if (M < 1 || K < 1) return 0;
  if (scheme == PARADIS_GEMM_F16X2) return (size_t)split_image_chunks(M, K, 2) * 16 + F16_TAIL_BYTES;
  if (scheme == PARADIS_GEMM_BF16) return (size_t)split_image_chunks(M, K, 1) * 16;
  return scheme == PARADIS_GEMM_BF16X3 ? (size_t)split_image_chunks(M, K, 3) * 16 : 0;
}

extern "C" int paradis_amax_partials(const float* x, int B, int64_t inner, int64_t bs, uint32_t* partials,
                                     void* stream) {
  PD_REQUIRE(partials != nullptr && B >= 0 && inner >= 0 && (x != nullptr || B == 0 || inner == 0),
             "amax_partials: bad arguments");
  launch_amax(x, B, inner, bs, partials, (hipStream_t)stream);
  PD_CHECK_LAUNCH("amax_partials");
  return 0;
}

// Split image (tile order) of A = W[M,K] (transpose = 0) or of A = W^T[K,M] (transpose = 1, from the same
// row-major W[M,K]); out holds split_bytes(M,K,scheme) resp. split_bytes(K,M,scheme).  BF16X3: h/m/l bf16
// planes.  F16X2: h/l f16 planes of W 2^e and, behind them, the bits of max |W|.
extern "C" int paradis_pw_gemm_split_weights(const float* W, int M, int K, int transpose, int scheme, void* out,
                                             void* stream) {
  PD_REQUIRE(W != nullptr && out != nullptr && M >= 1 && K >= 1, "pw_gemm_split_weights: bad arguments");
  PD_REQUIRE(scheme == PARADIS_GEMM_BF16X3 || scheme == PARADIS_GEMM_F16X2 || scheme == PARADIS_GEMM_BF16,
             "pw_gemm_split_weights: unknown scheme %d", scheme);
  const int AM = transpose ? K : M, AK = transpose ? M : K;
  const int KT = split_image_ktiles(AK, scheme == PARADIS_GEMM_BF16 ? 1 : 3);
  const int64_t units = (int64_t)((AM + BM - 1) / BM) * KT * 256;
  const int blocks = (int)std::min<int64_t>((units + 255) / 256, 4096);
  if (scheme == PARADIS_GEMM_F16X2) {
    uint32_t* tail = reinterpret_cast<uint32_t*>((char*)out + (size_t)split_image_chunks(AM, AK, 2) * 16);
    launch_amax(W, 1, (int64_t)M * K, 0, tail + 4, (hipStream_t)stream);
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W,
                       (int64_t)(transpose ? 1 : K), (int64_t)(transpose ? K : 1), AM, AK, KT, units, (u32x4*)out, tail);
  } else if (scheme == PARADIS_GEMM_BF16) {
    hipLaunchKernelGGL(split_weights_kernel<1>, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, W,
                       (int64_t)(transpose ? 1 : K), (int64_t)(transpose ? K : 1), AM, AK, KT, units,
                       (int64_t)0, (int64_t)0, (u32x4*)out);
  } else {
    hipLaunchKernelGGL(split_weights_kernel<3>, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, W,
                       (int64_t)(transpose ? 1 : K), (int64_t)(transpose ? K : 1), AM, AK, KT, units,
                       (int64_t)0, (int64_t)0, (u32x4*)out);
  }
  PD_CHECK_LAUNCH("pw_gemm_split_weights");
  return 0;
}

// images of W[M,K] (-> out, split_bytes(M,K,scheme)) and of W^T (-> out_t, split_bytes(K,M,scheme)) in one launch:
// PARADIS_GEMM_BF16X3 or PARADIS_GEMM_BF16 (the f16x2 image needs the amax of W first: paradis_pw_gemm_split_weights)
extern "C" int paradis_pw_gemm_split_weights_pair_scheme(const float* W, int M, int K, int scheme, void* out, void* out_t,
                                                         void* stream) {
  PD_REQUIRE(W != nullptr && out != nullptr && out_t != nullptr && out != out_t && M >= 1 && K >= 1,
             "pw_gemm_split_weights_pair: bad arguments");
  PD_REQUIRE(scheme == PARADIS_GEMM_BF16X3 || scheme == PARADIS_GEMM_BF16,
             "pw_gemm_split_weights_pair: scheme %d has no paired images", scheme);
  const int np = scheme == PARADIS_GEMM_BF16 ? 1 : 3;
  const int KT = split_image_ktiles(K, np), KTt = split_image_ktiles(M, np);      // (bf16-mixed images: an even count of k16 tiles)
  const int64_t units = (int64_t)((M + BM - 1) / BM) * KT * 256, units_t = (int64_t)((K + BM - 1) / BM) * KTt * 256;
  const int blocks = (int)std::min<int64_t>((std::max(units, units_t) + 255) / 256, 4096);
  if (scheme == PARADIS_GEMM_BF16X3)
    hipLaunchKernelGGL(split_weights_pair_kernel<3>, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, W, M, K, KT, KTt,
                       units, units_t, (u32x4*)out, (u32x4*)out_t);
  else
    hipLaunchKernelGGL(split_weights_pair_kernel<1>, dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, W, M, K, KT, KTt,
                       units, units_t, (u32x4*)out, (u32x4*)out_t);
  PD_CHECK_LAUNCH("pw_gemm_split_weights_pair");
  return 0;
}
// (ABI 7's spelling: the bf16x3 pair)
extern "C" int paradis_pw_gemm_split_weights_pair(const float* W, int M, int K, void* out, void* out_t, void* stream) {
  return paradis_pw_gemm_split_weights_pair_scheme(W, M, K, PARADIS_GEMM_BF16X3, out, out_t, stream);
}

namespace {
// A = split image at `img` of an [AM, AK] matrix; F16X2 needs the activations' amax partials
int run_split(GemmArgs d, const void* img, int AM, int AK, int scheme, const uint32_t* b_amax, const char* what,
              hipStream_t st) {
  d.A = (const float*)img; d.a_bs = 0;
  if (scheme == PARADIS_GEMM_F16X2) {
    if (b_amax == nullptr) { paradis_set_error(what); return 1; }
    d.a_amax = reinterpret_cast<const uint32_t*>((const char*)img + (size_t)split_image_chunks(AM, AK, 2) * 16);
    d.b_amax = b_amax;
  }
  return launch_split(d, scheme, st);
}
// bf16-stored tensors exist in the bf16-mixed scheme only; the DMA'd activation operand needs whole 16-byte chunks
int check_io16(const char* name, int io16, int scheme, const void* Bop, int64_t b_bs, int N, bool has_res) {
  if (io16 == 0) return 0;
  PD_REQUIRE(scheme == PARADIS_GEMM_BF16, "%s: bf16-stored tensors need the PARADIS_GEMM_BF16 scheme", name);
  PD_REQUIRE((io16 & ~(IO_B16 | IO_C16 | IO_ZM16 | IO_A16)) == 0, "%s: unknown io16 bits %d", name, io16);
  PD_REQUIRE(!(io16 & IO_C16) || !has_res, "%s: a bf16 output cannot carry the fp32 residual", name);
  if (io16 & IO_B16)
    PD_REQUIRE(N % 8 == 0 && N >= 8 && b_bs % 8 == 0 && (reinterpret_cast<uintptr_t>(Bop) & 15) == 0,
               "%s: a bf16 activation operand needs N %% 8 == 0 and 16-byte aligned planes", name);
  return 0;
}
}  // namespace

static int pw_gemm_fwd_impl(const float* Wt, const float* WtT, const void* Wsplit, int scheme,
                            const uint32_t* x_amax, const float* X,
                            const float* bias, const float* map, const float* m8,
                            const float* pwT, int cin, const float* res, const float* gate, float* Y, float* zpre,
                            int B, int M, int K, int N, int64_t x_bs, int64_t res_bs,
                            int64_t y_bs, int act, void* stream, int io16 = 0) {
  if (int e = check_gemm("pw_gemm_fwd", B, M, K, N)) return e;
  if (int e = check_io16("pw_gemm_fwd", io16, scheme, X, x_bs, N, res != nullptr)) return e;
  PD_REQUIRE(gate == nullptr || res != nullptr || B == 0, "pw_gemm_fwd: a gate needs the tensor it blends with (res)");
  PD_REQUIRE(act >= 0 && act <= 2, "pw_gemm_fwd: unknown activation code %d", act);
  PD_REQUIRE(known_scheme(scheme) && (Wsplit != nullptr) == (scheme != PARADIS_GEMM_EXACT),
             "pw_gemm_fwd: scheme %d needs %s weight image", scheme, scheme ? "a" : "no");
  PD_REQUIRE((m8 == nullptr) == (pwT == nullptr) && (pwT == nullptr || (cin >= 1 && M % 4 == 0)),
             "pw_gemm_fwd: projected bias needs m8, pwT, cin >= 1 and M %% 4 == 0");
  if (B == 0) return 0;
  GemmArgs g{};
  g.m8 = m8; g.pw = pwT; g.cin = cin;
  g.A = Wt; g.B = X; g.C = Y; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = N; g.ldc = N;
  g.a_bs = 0; g.b_bs = x_bs; g.c_bs = y_bs; g.nbatch = B; g.inner = 0;
  g.bias = bias; g.map = map; g.res = res; g.res_bs = res_bs; g.zmul = nullptr; g.zout = zpre;
  g.zout_bs = (int64_t)M * N; g.act = act; g.gate = gate;
  g.stagger = g_stagger;
  g.io16 = io16;
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * B;
  if (Wsplit != nullptr) {   // split image of the weights: split kernel (any shape)
    if (int e = run_split(g, Wsplit, M, K, scheme, x_amax, "pw_gemm_fwd: the f16x2 scheme needs x_amax",
                          (hipStream_t)stream)) return e;
    PD_CHECK_LAUNCH("pw_gemm_fwd(split)");
    return 0;
  }
  if (WtT != nullptr) {   // weights also supplied as [K,M]: row-contiguous A operand -> LDS-DMA kernel
    GemmArgs d = g;
    d.A = WtT; d.lda = M;
    if (dma_eligible(d)) {
      launch_gemm_dma(d, grid, (hipStream_t)stream);
      PD_CHECK_LAUNCH("pw_gemm_fwd(dma)");
      return 0;
    }
  }
  if (int e = launch_gemm<true, false>(g, grid, (hipStream_t)stream)) return e;
  PD_CHECK_LAUNCH("pw_gemm_fwd");
  return 0;
}

extern "C" int paradis_pw_gemm_fwd(const float* Wt, const float* WtT, const void* Wsplit, int scheme,
                                   const uint32_t* x_amax, const float* X,
                                   const float* bias, const float* map, const float* m8,
                                   const float* pwT, int cin, const float* res, float* Y, float* zpre,
                                   int B, int M, int K, int N, int64_t x_bs, int64_t res_bs,
                                   int64_t y_bs, int act, void* stream) {
  return pw_gemm_fwd_impl(Wt, WtT, Wsplit, scheme, x_amax, X, bias, map, m8, pwT, cin, res, nullptr, Y, zpre, B, M, K, N,
                          x_bs, res_bs, y_bs, act, stream);
}

// Y = res + sigmoid(gate[m]) (act(W X + ...) - res): the gated blend of the advected field with the field it was
// advected from (reference model/paradis.py:239-243) inside the epilogue of the up-projection's last layer - the
// advected tensor is never written.  Same bits as paradis_pw_gemm_fwd followed by paradis_gated_blend_fwd.
extern "C" int paradis_pw_gemm_fwd_gated(const float* Wt, const float* WtT, const void* Wsplit, int scheme,
                                         const uint32_t* x_amax, const float* X,
                                         const float* bias, const float* map, const float* m8,
                                         const float* pwT, int cin, const float* res, const float* gate, float* Y,
                                         float* zpre, int B, int M, int K, int N, int64_t x_bs, int64_t res_bs,
                                         int64_t y_bs, int act, void* stream) {
  PD_REQUIRE(gate != nullptr && (res != nullptr || B == 0), "pw_gemm_fwd_gated: gate [M] and res required");
  return pw_gemm_fwd_impl(Wt, WtT, Wsplit, scheme, x_amax, X, bias, map, m8, pwT, cin, res, gate, Y, zpre, B, M, K, N,
                          x_bs, res_bs, y_bs, act, stream);
}

// PARADIS_GEMM_BF16 with bf16-STORED tensors (round 6): io16 = PARADIS_IO_X16 (X is bf16 [B][K,N]) | PARADIS_IO_Y16 (Y and
// zpre are written as bf16; no residual then).  Everything else as paradis_pw_gemm_fwd / _gated (gate may be NULL).
extern "C" int paradis_pw_gemm_fwd16(const void* Wsplit, const void* X, const float* bias, const float* map,
                                     const float* m8, const float* pwT, int cin, const float* res, const float* gate,
                                     void* Y, void* zpre, int B, int M, int K, int N, int64_t x_bs, int64_t res_bs,
                                     int64_t y_bs, int act, int io16, void* stream) {
  PD_REQUIRE((io16 & ~(IO_B16 | IO_C16)) == 0, "pw_gemm_fwd16: io16 may name X (1) and Y (2)");
  PD_REQUIRE(Wsplit != nullptr, "pw_gemm_fwd16: weight image required");
  return pw_gemm_fwd_impl(nullptr, nullptr, Wsplit, PARADIS_GEMM_BF16, nullptr, (const float*)X, bias, map, m8, pwT, cin,
                          res, gate, (float*)Y, (float*)zpre, B, M, K, N, x_bs, res_bs, y_bs, act, stream, io16);
}

// Plain batched GEMM  C_b[M,N] = A_b[M,K] B_b[K,N]  (row-major, batch strides in elements) on the same
// kernels; AT (optional) = the [K,M] transposes of A_b with batch stride at_bs, which lets the LDS-DMA
// kernel run; split_ws (optional, nbatch * paradis_pw_gemm_split_bytes(M,K) bytes) selects the bf16-split
// arithmetic instead (the A matrices are split into it first).  Used by the Newton-Schulz iteration of the Muon step (muon.hip): same-shape weight
// matrices are stacked so that one launch fills the chip (a single 896^3 product is 49 tiles).
extern "C" int paradis_bgemm(const float* A, const float* AT, const float* Bm, float* C, int nbatch, int M,
                             int K, int N, int64_t a_bs, int64_t at_bs, int64_t b_bs, int64_t c_bs,
                             void* split_ws, void* stream) {
  PD_REQUIRE(nbatch >= 0 && M >= 1 && K >= 1 && N >= 1, "bgemm: bad shape");
  if (nbatch == 0) return 0;
  GemmArgs g{};
  g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K;
  g.lda = K; g.ldb = N; g.ldc = N;
  g.a_bs = a_bs; g.b_bs = b_bs; g.c_bs = c_bs; g.nbatch = nbatch; g.inner = 0;
  g.stagger = g_stagger;
  const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * nbatch;
  PD_REQUIRE(tiles < (1ll << 31), "bgemm: too many tiles");
  const int grid = (int)tiles;
  if (split_ws != nullptr) {   // bf16-split path: images of the nbatch A matrices in split_ws
    PD_REQUIRE(nbatch <= 65535, "bgemm: too many batches for the split path");
    const int KT = (K + SBK - 1) / SBK;
    const int64_t chunks = split_image_chunks(M, K), units = (int64_t)((M + BM - 1) / BM) * KT * 256;
    const int blocks = (int)std::min<int64_t>((units + 255) / 256, 1024);
    hipLaunchKernelGGL(split_weights_kernel<3>, dim3(blocks, nbatch), dim3(256), 0, (hipStream_t)stream, A,
                       (int64_t)K, (int64_t)1, M, K, KT, units, a_bs, chunks, (u32x4*)split_ws);
    GemmArgs d = g;
    d.A = (const float*)split_ws; d.a_bs = chunks;
    if (int e = launch_split(d, PARADIS_GEMM_BF16X3, (hipStream_t)stream)) return e;
    PD_CHECK_LAUNCH("bgemm(split)");
    return 0;
  }
  if (AT != nullptr) {
    GemmArgs d = g;
    d.A = AT; d.lda = M; d.a_bs = at_bs;
    if (dma_eligible(d)) {
      launch_gemm_dma(d, grid, (hipStream_t)stream);
      PD_CHECK_LAUNCH("bgemm(dma)");
      return 0;
    }
  }
  if (int e = launch_gemm<true, false>(g, grid, (hipStream_t)stream)) return e;
  PD_CHECK_LAUNCH("bgemm");
  return 0;
}

static int pw_gemm_dgrad_impl(const float* Wt, const void* WTsplit, int scheme, const uint32_t* dy_amax,
                              const float* dY, const float* zpre,
                              const float* addend, float* dX, int B, int M, int K, int N,
                              int64_t dy_bs, int64_t z_bs, int64_t add_bs, int64_t dx_bs,
                              int act, void* stream, int io16) {
  // W is [M,K] (M = Co, K = Ci); result dX is [K,N] per sample: GEMM with M' = K, K' = M.
  if (int e = check_gemm("pw_gemm_dgrad", B, K, M, N)) return e;
  if (int e = check_io16("pw_gemm_dgrad", io16, scheme, dY, dy_bs, N, addend != nullptr)) return e;
  PD_REQUIRE(act >= 0 && act <= 2, "pw_gemm_dgrad: unknown activation code %d", act);
  PD_REQUIRE(known_scheme(scheme) && (WTsplit != nullptr) == (scheme != PARADIS_GEMM_EXACT),
             "pw_gemm_dgrad: scheme %d needs %s weight image", scheme, scheme ? "a" : "no");
  if (B == 0) return 0;
  GemmArgs g{};
  g.A = Wt; g.B = dY; g.C = dX; g.M = K; g.N = N; g.K = M;
  g.lda = K; g.ldb = N; g.ldc = N;
  g.a_bs = 0; g.b_bs = dy_bs; g.c_bs = dx_bs; g.nbatch = B; g.inner = 0;
  g.res = addend; g.res_bs = add_bs; g.zmul = zpre; g.zmul_bs = z_bs; g.act = zpre ? act : 0;
  g.stagger = g_stagger;
  g.io16 = io16;
  const int grid = ((K + BM - 1) / BM) * ((N + BN - 1) / BN) * B;
  if (WTsplit != nullptr) {   // split image of W^T
    if (int e = run_split(g, WTsplit, K, M, scheme, dy_amax, "pw_gemm_dgrad: the f16x2 scheme needs dy_amax",
                          (hipStream_t)stream)) return e;
    PD_CHECK_LAUNCH("pw_gemm_dgrad(split)");
    return 0;
  }
  if (dma_eligible(g)) {
    launch_gemm_dma(g, grid, (hipStream_t)stream);
    PD_CHECK_LAUNCH("pw_gemm_dgrad(dma)");
    return 0;
  }
  if (int e = launch_gemm<false, false>(g, grid, (hipStream_t)stream)) return e;
  PD_CHECK_LAUNCH("pw_gemm_dgrad");
  return 0;
}

extern "C" int paradis_pw_gemm_dgrad(const float* Wt, const void* WTsplit, int scheme, const uint32_t* dy_amax,
                                     const float* dY, const float* zpre,
                                     const float* addend, float* dX, int B, int M, int K, int N,
                                     int64_t dy_bs, int64_t z_bs, int64_t add_bs, int64_t dx_bs,
                                     int act, void* stream) {
  return pw_gemm_dgrad_impl(Wt, WTsplit, scheme, dy_amax, dY, zpre, addend, dX, B, M, K, N, dy_bs, z_bs, add_bs, dx_bs,
                            act, stream, 0);
}

// PARADIS_GEMM_BF16 with bf16-STORED tensors: io16 = PARADIS_IO_X16 (dY is bf16) | PARADIS_IO_Y16 (dX is written as bf16)
// | PARADIS_IO_Z16 (zpre is bf16).
extern "C" int paradis_pw_gemm_dgrad16(const void* WTsplit, const void* dY, const void* zpre, void* dX, int B, int M,
                                       int K, int N, int64_t dy_bs, int64_t z_bs, int64_t dx_bs, int act, int io16,
                                       void* stream) {
  PD_REQUIRE((io16 & ~(IO_B16 | IO_C16 | IO_ZM16)) == 0, "pw_gemm_dgrad16: io16 may name dY (1), dX (2) and zpre (4)");
  PD_REQUIRE(WTsplit != nullptr, "pw_gemm_dgrad16: weight image required");
  return pw_gemm_dgrad_impl(nullptr, WTsplit, PARADIS_GEMM_BF16, nullptr, (const float*)dY, (const float*)zpre, nullptr,
                            (float*)dX, B, M, K, N, dy_bs, z_bs, 0, dx_bs, act, stream, io16);
}

extern "C" size_t paradis_pw_gemm_wgrad_ws_bytes(int B, int M, int K, int N) {
  const int b = std::max(B, 1);
  const int S = std::max({wgrad_splits(b, M, K, N, DBK, wgrad_dma_wgs()), wgrad_splits(b, M, K, N, g_bk, g_wg_per_cu),
                          wgrad_splits(b, M, K, N, SBK, 3), wgrad_splits_tall(b, M, K, N),
                          wgrad_splits_square(b, M, K, N)});
  return (size_t)S * M * ((size_t)K + 1) * sizeof(float) + 256;   // slabs + row-sum partials
}

// K-range slabs the split weight-gradient kernel runs for this shape (1, or an even number: see wgrad_splits)
extern "C" int paradis_pw_gemm_wgrad_slabs(int B, int M, int K, int N) {
  if (M < 1 || K < 1 || N < 1) return 0;
  return wgrad_splits(std::max(B, 1), M, K, N, SBK, 3);
}

extern "C" int paradis_bias_grads(const float* dz, float* gmap, float* gbias, int B, int C, int P,
                                  int64_t dz_bs, void* stream);

static int pw_gemm_wgrad_impl(const float* dY, const float* X, float* dW, float* gbias, int B,
                              int M, int K, int N, int64_t dy_bs, int64_t x_bs, int scheme,
                              const uint32_t* dy_amax, const uint32_t* x_amax, void* workspace,
                              void* stream, int io16) {
  // dW[M,K] = sum_b dY[b][M,N] . X[b][K,N]^T : GEMM with M'=M, N'=K, K'=N, reduced over samples.
  if (int e = check_gemm("pw_gemm_wgrad", 1, M, N, K)) return e;
  if (io16) {
    PD_REQUIRE(scheme == PARADIS_GEMM_BF16 && (io16 & ~(IO_A16 | IO_B16)) == 0,
               "pw_gemm_wgrad: bf16-stored operands need the PARADIS_GEMM_BF16 scheme (io16 = dY 8 | X 1)");
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    PD_REQUIRE(N % SBK == 0 && a16(dY) && a16(X) && dy_bs % ((io16 & IO_A16) ? 8 : 4) == 0 &&
               x_bs % ((io16 & IO_B16) ? 8 : 4) == 0,
               "pw_gemm_wgrad: bf16-stored operands need N %% 16 == 0 and 16-byte aligned rows");
  }
  PD_REQUIRE(known_scheme(scheme), "pw_gemm_wgrad: unknown scheme %d", scheme);
  PD_REQUIRE(scheme != PARADIS_GEMM_F16X2 || (dy_amax != nullptr && x_amax != nullptr),
             "pw_gemm_wgrad: the f16x2 scheme needs dy_amax and x_amax");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (pd_zero_async(dW, (size_t)M * K * sizeof(float), st) != hipSuccess) return 2;
    if (gbias && pd_zero_async(gbias, (size_t)M * sizeof(float), st) != hipSuccess) return 2;
    return 0;
  }
  // a split scheme: both operands are split in registers (same layout requirements as the LDS-DMA kernel)
  const bool use_split = io16 != 0 || (scheme != PARADIS_GEMM_EXACT && wgrad_vec_layout(N, dy_bs, x_bs, dY, X));
  const bool dma = use_split || wgrad_dma_ok(N, dy_bs, x_bs, dY, X);   // "dma" = kernels with fused row sums
  const bool tall = use_split && scheme == PARADIS_GEMM_BF16 && wgrad_tall_ok(M);
  const bool square = tall && wgrad_square_on() && ((K + 255) / 256) * 256 * 7 <= K * 8;
  const int S = square ? wgrad_splits_square(B, M, K, N) : tall ? wgrad_splits_tall(B, M, K, N)
              : use_split ? wgrad_splits(B, M, K, N, SBK, 3)
                          : dma ? wgrad_splits(B, M, K, N, DBK, wgrad_dma_wgs())
                                : wgrad_splits(B, M, K, N, g_bk, g_wg_per_cu);
  PD_REQUIRE(workspace != nullptr, "pw_gemm_wgrad: workspace required");
  float* rowsum_ws = (float*)workspace + (size_t)S * M * K;   // [S][M], behind the slabs
  if (gbias && !dma) {   // register-staged kernel has no fused row sums: separate reduction pass
    if (int e = paradis_bias_grads(dY, nullptr, gbias, B, M, N, dy_bs, stream)) return e;
  }
  GemmArgs g{};
  g.A = dY; g.B = X; g.C = S > 1 ? (float*)workspace : dW;
  g.M = M; g.N = K; g.K = N;
  g.lda = N; g.ldb = N; g.ldc = K;
  g.a_bs = 0; g.b_bs = 0; g.c_bs = (int64_t)M * K; g.nbatch = S;
  g.inner = B; g.a_is = dy_bs; g.b_is = x_bs;
  g.stagger = g_stagger;
  g.rowsum = (gbias && dma) ? rowsum_ws : nullptr;
  const int grid = (tall ? (M + 255) / 256 : (M + BM - 1) / BM) * ((K + BN - 1) / BN) * S;
  if (square || tall) {
    const int g2 = square ? ((M + 255) / 256) * ((K + 255) / 256) * S : grid;
    if (int e = pd_amp_launch_wgrad(g, io16, square ? 3 : 2, g2, st)) return e;
  } else if (use_split && scheme == PARADIS_GEMM_F16X2) {
    g.a_amax = dy_amax; g.b_amax = x_amax;
    hipLaunchKernelGGL(pw_gemm_wgrad_split_kernel<2>, dim3(grid), dim3(256), split_lds_wgrad(2), st, g);
  } else if (use_split && scheme == PARADIS_GEMM_BF16) {
    if (int e = pd_amp_launch_wgrad(g, io16, io16 ? 1 : 0, grid, st)) return e;
  } else if (use_split) {
    hipLaunchKernelGGL(pw_gemm_wgrad_split_kernel<3>, dim3(grid), dim3(256), split_lds_wgrad(3), st, g);
  } else if (dma) {
    const size_t bytes = (size_t)g_wgrad_dma_stages * 2 * DTILE * sizeof(float);
    if (g_wgrad_dma_stages == 2)
      hipLaunchKernelGGL(pw_gemm_wgrad_dma_kernel<2>, dim3(grid), dim3(256), bytes, st, g);
    else
      hipLaunchKernelGGL(pw_gemm_wgrad_dma_kernel<3>, dim3(grid), dim3(256), (size_t)3 * 2 * DTILE * sizeof(float), st, g);
  } else if (int e = launch_gemm<true, true>(g, grid, st)) return e;
  {
    // slab sums in a fixed order; with one slab the GEMM wrote dW itself and only the row sums (if any) remain
    const int64_t n = S > 1 ? (int64_t)M * K : 0;
    const int vec = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(dW)) & 15) == 0;
    const int blocks1 = n ? (int)std::min<int64_t>(((vec ? n / 4 : n) + 255) / 256, 2048) : 0;
    const int n2 = g.rowsum ? M : 0, blocks2 = (n2 + 255) / 256;
    if (blocks1 + blocks2 > 0)
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks1 + blocks2), dim3(256), 0, st, (const float*)workspace, dW, n, S,
                         vec, (const float*)rowsum_ws, gbias, n2, blocks1);
  }
  PD_CHECK_LAUNCH("pw_gemm_wgrad");
  return 0;
}

extern "C" int paradis_pw_gemm_wgrad(const float* dY, const float* X, float* dW, float* gbias, int B,
                                     int M, int K, int N, int64_t dy_bs, int64_t x_bs, int scheme,
                                     const uint32_t* dy_amax, const uint32_t* x_amax, void* workspace,
                                     void* stream) {
  return pw_gemm_wgrad_impl(dY, X, dW, gbias, B, M, K, N, dy_bs, x_bs, scheme, dy_amax, x_amax, workspace, stream, 0);
}

// PARADIS_GEMM_BF16 with bf16-STORED operands: io16 = PARADIS_IO_DY16 (dY is bf16) | PARADIS_IO_X16 (X is bf16); dW and the
// bias gradient stay fp32.  Workspace: paradis_pw_gemm_wgrad_ws_bytes.
extern "C" int paradis_pw_gemm_wgrad16(const void* dY, const void* X, float* dW, float* gbias, int B, int M, int K, int N,
                                       int64_t dy_bs, int64_t x_bs, int io16, void* workspace, void* stream) {
  return pw_gemm_wgrad_impl((const float*)dY, (const float*)X, dW, gbias, B, M, K, N, dy_bs, x_bs, PARADIS_GEMM_BF16,
                            nullptr, nullptr, workspace, stream, io16);
}
#endif   // GEMM_PART < 2
