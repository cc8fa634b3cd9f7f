/*
 * paradis_hip.h -- C ABI of libparadis_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the PARADIS advection-diffusion-reaction hot path.
 * The reference has no FFI layer (it is pure Python on ATen); each entry point
 * below replaces the ATen call sequence of the cited reference lines and is
 * what a ctypes binding in the reference's model/ package would call (see
 * INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (unless noted), NCHW, planes
 *     (H*W) contiguous; `*_bs` arguments are batch strides in ELEMENTS so that
 *     channel slices of a larger tensor can be read/written without copies;
 *   - inputs are borrowed, outputs are caller-allocated, no global mutable
 *     state, no host synchronisation; kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*), so calls are graph-capturable;
 *   - return 0 on success; non-zero = rejected arguments (1) or HIP launch
 *     failure (2); paradis_last_error() returns a thread-local message;
 *   - `act`: 0 none, 1 SiLU, 2 GELU(erf).  `mode`: 1 bilinear, 2 bicubic.
 */
#ifndef PARADIS_HIP_H
#define PARADIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARADIS_ACT_NONE 0
#define PARADIS_ACT_SILU 1
#define PARADIS_ACT_GELU 2
#define PARADIS_INTERP_BILINEAR 1
#define PARADIS_INTERP_BICUBIC 2
/* `flags` of paradis_sl_advect_{fwd,bwd}.  0 = automatic schedule: the whole padded plane in LDS when it
 * fits 64 KiB (one wave per latitude row when W == 64 and PARADIS_ADVECT_SEPARABLE is set), otherwise
 * a windowed schedule with an L2 path for taps outside the window: with PARADIS_ADVECT_SEPARABLE 128-column strips
 * walked top to bottom through a ring of padded rows in LDS (round 4), otherwise 128-column tiles with a halo.
 *   bit 0  PARADIS_ADVECT_GENERIC   whole-plane schedule with per-point table loads even when W == 64
 *   bit 1  PARADIS_ADVECT_TILED     windowed schedule regardless of the plane size
 *   bit 2  PARADIS_ADVECT_SEPARABLE the caller vouches that sin_lat/cos_lat are constant along a row and
 *                                   lon along a column (every regular lat-lon grid)
 *   bit 3  PARADIS_ADVECT_TILES     diagnostic: the 64 x 128 / 16 x 128 tile schedule of rounds 2-3 instead of strips
 *   bit 4  PARADIS_ADVECT_STRIPS    backward: 128-column strips even where the full-circle ring (W <= 256) would run
 *   bits 8-15  window halo in longitude of the windowed schedules (0 = built-in default, else halo + 1)
 *   bits 16-23 the same for the backward kernel only */
#define PARADIS_ADVECT_AUTO 0
#define PARADIS_ADVECT_GENERIC 1
#define PARADIS_ADVECT_TILED 2
#define PARADIS_ADVECT_SEPARABLE 4
#define PARADIS_ADVECT_TILES 8
#define PARADIS_ADVECT_STRIPS 16
#define PARADIS_ADVECT_HALO_SHIFT 8
#define PARADIS_ADVECT_HALO_BWD_SHIFT 16
#define PARADIS_ADVECT_HALO(h) (((h) + 1) << PARADIS_ADVECT_HALO_SHIFT)

int paradis_abi_version(void);   /* 9: bf16-stored tensors of the bf16-mixed mode (paradis_pw_gemm_fwd16 / _dgrad16 / _wgrad16, paradis_bias_grads16, paradis_channel_norm_fwd16 / _bwd16, paradis_dwconv_geo_fwd16 / _bwd16, paradis_act_bwd16): additions only; 8: PARADIS_GEMM_BF16 scheme, paradis_pw_gemm_wgrad_slabs (query); no signature changed; 7: paradis_dwconv_geo_dgrad_add, paradis_dwconv_geo_bwd, paradis_pw_gemm_split_weights_pair, paradis_pw_gemm_fwd_gated, paradis_gated_blend_bwd_out; 6: paradis_sl_advect_ws_bytes takes the call's flags (strip schedule); 5: no amax side outputs, lat_cells table of sl_advect_* (4: GEMM `scheme` arguments, paradis_amax_partials; 3: `flags` of sl_advect_*) */
const char* paradis_last_error(void);

/* ---- a1: GeoCyclicPadding.forward (reference model/padding.py:11-39) and its adjoint.
 * x [planes,H,W] -> y [planes,H+2p,W+2p];  gx[src(i,j)] += gy[i,j]. Integer index map, bit-exact. */
int paradis_geocyclic_pad_fwd(const float* x, float* y, int64_t planes, int H, int W, int p, void* stream);
int paradis_geocyclic_pad_bwd(const float* gy, float* gx, int64_t planes, int H, int W, int p, void* stream);

/* ---- a3-a5: fused core of NeuralSemiLagrangian.forward (reference model/advection.py:129-169):
 * pole mean -> departure point (advection.py:74-98) -> virtual geocyclic index map ->
 * bilinear/bicubic gather (ATen grid_sampler_2d semantics, align_corners, zeros) -> pole mean.
 * field/out [B,K,H,W]; u,v [B,K,H,W] with batch stride uv_bs; sin_lat/cos_lat/lon tables [H*W].
 * lat_cells [H*W] (optional, NULL allowed): the arrival latitude in PADDED cells,
 *   p + (lat - min_lat) (H-1)/d_lat   with p = 2 (bicubic) | 1 (bilinear), evaluated in double and rounded once.
 *   The separable schedules (PARADIS_ADVECT_SEPARABLE) need it: they take the departure latitude relative to the
 *   arrival latitude at small displacements (no asin half-angle chain, no 1/cos(lat) amplification of the rounding
 *   of sin(lat_d) next to the poles); without the table the per-point kernels run. */
int paradis_sl_advect_fwd(const float* field, const float* u, const float* v, float* out,
                          const float* sin_lat, const float* cos_lat, const float* lat_cells, const float* lon,
                          int B, int K, int H, int W, int64_t f_bs, int64_t uv_bs, int64_t o_bs,
                          float dt, float min_lat, float min_lon, float d_lat, float d_lon,
                          int mode, int flags, void* workspace, void* stream);
/* gfield [B,K,H,W] (batch stride gf_bs), gu/gv with batch stride guv_bs.
 * workspace (fwd and bwd): >= paradis_sl_advect_ws_bytes(B,K,H,W,flags) bytes for the `flags` of the call (with
 * PARADIS_DETERMINISTIC=1 this includes the 64-bit integer plane the windowed backward accumulates into: 8 bytes per
 * gather point; for the strip schedule it includes the lists of points whose taps leave the window: 12 bytes per gather
 * point). */
size_t paradis_sl_advect_ws_bytes(int B, int K, int H, int W, int flags);
int paradis_sl_advect_bwd(const float* gout, const float* field, const float* u, const float* v,
                          float* gfield, float* gu, float* gv,
                          const float* sin_lat, const float* cos_lat, const float* lat_cells, const float* lon,
                          int B, int K, int H, int W, int64_t go_bs, int64_t f_bs, int64_t uv_bs,
                          int64_t gf_bs, int64_t guv_bs,
                          float dt, float min_lat, float min_lon, float d_lat, float d_lon,
                          int mode, int flags, void* workspace, void* stream);

/* ---- a7 (depthwise half of SepConv, reference model/blocks.py:101-113) and the static
 * encoder's GeoCyclicPadding(3)+Conv2d(groups=C) (reference model/paradis.py:189-190):
 * k x k per-channel stencil on the virtual geocyclic halo. w [C,k,k]; bias [C] or NULL; k odd, 1..11. */
int paradis_dwconv_geo_fwd(const float* x, const float* w, const float* bias, float* y,
                           int B, int C, int H, int W, int k, void* stream);
int paradis_dwconv_geo_dgrad(const float* gy, const float* w, float* gx,
                             int B, int C, int H, int W, int k, void* stream);
/* gx = dgrad(gy) + addend: the stencil's input has a second consumer (the block input that the gated blend of
 * reference model/paradis.py:239-240 reads next to the advection's down-projection); its gradient enters here
 * instead of through an accumulation pass.  addend [B,C,H,W], must not alias gx.  (ABI 7) */
int paradis_dwconv_geo_dgrad_add(const float* gy, const float* w, const float* addend, float* gx,
                                 int B, int C, int H, int W, int k, void* stream);
/* Both gradients from one call (autograd of the same module): gx = dgrad(gy) (+ addend, nullable), gw [C,k,k],
 * gbias [C] or NULL.  k = 5 with 16-byte aligned tensors and W = 64, H <= 32 (one tile) or H >= 32, W >= 64, W % 4 == 0
 * (several tiles): one kernel that reads gy once; otherwise the two kernels above.  Bit-identical to them.
 * workspace: paradis_dwconv_geo_wgrad_ws_bytes.  (ABI 7) */
int paradis_dwconv_geo_bwd(const float* gy, const float* x, const float* w, const float* addend, float* gx,
                           float* gw, float* gbias, int B, int C, int H, int W, int k, void* workspace, void* stream);
size_t paradis_dwconv_geo_wgrad_ws_bytes(int B, int C, int H, int W, int k);
int paradis_dwconv_geo_wgrad(const float* gy, const float* x, float* gw, float* gbias,
                             int B, int C, int H, int W, int k, void* workspace, void* stream);

/* ---- a11: PhysicalDownsample (reference model/blocks.py:57-71): geocyclic 5x5 box mean, stride s */
int paradis_avgpool_geo_fwd(const float* x, float* y, int64_t planes, int H, int W, int stride, void* stream);
int paradis_avgpool_geo_bwd(const float* gy, float* gx, int64_t planes, int H, int W, int stride, void* stream);

/* ---- a14: Paradis.upsample (reference model/paradis.py:208-220): lon-periodic bilinear, align_corners */
int paradis_upsample_lonp_fwd(const float* x, float* y, int64_t planes, int Hc, int Wc, int H, int W, void* stream);
int paradis_upsample_lonp_bwd(const float* gy, float* gx, int64_t planes, int Hc, int Wc, int H, int W, void* stream);

/* ---- a6: CLinear / pointwise half of SepConv (reference model/blocks.py:86,110): per-sample GEMM.
 * Y[b] = epi( W[M,K] * X[b][K,N] ),  epi(v) = res + act(v + bias[m] + map[m,n]).
 * bias/map/res/zpre may be NULL.  zpre (if given) receives the pre-activation value.
 *
 * Three arithmetic schemes, all fp32 in / fp32 accumulate / fp32 out (`scheme` arguments):
 *   - PARADIS_GEMM_EXACT:  v_mfma_f32_32x32x2_f32 (an fmaf chain over k);
 *   - PARADIS_GEMM_BF16X3: each fp32 operand is decomposed exactly into three bf16 terms and the product is
 *     accumulated from the six partial products >= 2^-16 on v_mfma_f32_32x32x16_bf16 (relative
 *     truncation 2^-23 per product, fewer accumulation roundings: error vs fp64 not above the exact path's);
 *   - PARADIS_GEMM_F16X2 (opt-in; a block-exponent format, NOT the reference's per-element fp32 arithmetic):
 *     each operand TENSOR is scaled by a power of two that puts its largest magnitude
 *     into [2^14, 2^15) and every value is written as two f16 terms (22 significand bits; values more than
 *     ~2^17 below the tensor's maximum keep an absolute accuracy of 2^-39 max|x|); three partial products on
 *     v_mfma_f32_32x32x16_f16, result unscaled in the epilogue.  Error vs fp64 of a K = 1024 product: that of
 *     an fp32 SGEMM.  Needs the tensors' largest magnitudes: paradis_amax_partials for activations, the
 *     weight image carries its own.
 * fwd / dgrad take the split image of their weight operand (NULL with PARADIS_GEMM_EXACT).
 *
 * A fourth scheme is NOT fp32 arithmetic and exists for one purpose - the reference's shipped training mode
 * `use_amp: true` (config/paradis_settings.yaml:75 -> precision="bf16-mixed", train.py:56), in which torch.autocast runs
 * every conv2d with bf16 operands and a bf16 result:
 *   - PARADIS_GEMM_BF16 (ABI 8): operands rounded to bf16, ONE product on v_mfma_f32_32x32x16_bf16, fp32 accumulate;
 *     the pre-activation, the activated value and (dgrad) the activation-gradient product are rounded to bf16 where
 *     autocast's conv2d / activation round them, the residual / blend and the weight gradient stay fp32; tensors are
 *     stored as fp32.  The Python layer selects it only under torch.autocast(dtype=bfloat16). */
#define PARADIS_GEMM_EXACT 0
#define PARADIS_GEMM_BF16 1     /* ABI 8: the reference's bf16-mixed (AMP) mode, never the fp32 path: see below */
#define PARADIS_GEMM_F16X2 2
#define PARADIS_GEMM_BF16X3 3
#define PARADIS_AMAX_PARTIALS 1024
/* partials[PARADIS_AMAX_PARTIALS] <- bit patterns of partial maxima of |x| over B blocks of `inner`
 * contiguous floats (block stride bs); the tensor's maximum is the (unsigned) maximum of the words, a NaN
 * anywhere makes it a NaN pattern.  One read pass, no atomics, no pre-zeroing. */
int paradis_amax_partials(const float* x, int B, int64_t inner, int64_t bs, uint32_t* partials, void* stream);
size_t paradis_pw_gemm_split_bytes(int M, int K, int scheme);   /* bytes of the split image of an [M,K] A operand */
/* split image of A = W[M,K] (transpose 0; out: split_bytes(M,K,scheme)) or of A = W^T (transpose 1; out:
 * split_bytes(K,M,scheme)) from row-major W[M,K]; scheme = PARADIS_GEMM_BF16X3 or PARADIS_GEMM_F16X2 */
int paradis_pw_gemm_split_weights(const float* W, int M, int K, int transpose, int scheme, void* out, void* stream);
/* PARADIS_GEMM_BF16X3 images of W (-> out) and of W^T (-> out_t) in one launch: a training step needs both (ABI 7) */
int paradis_pw_gemm_split_weights_pair(const float* W, int M, int K, void* out, void* out_t, void* stream);
/* the same for PARADIS_GEMM_BF16X3 or PARADIS_GEMM_BF16 (ABI 9) */
int paradis_pw_gemm_split_weights_pair_scheme(const float* W, int M, int K, int scheme, void* out, void* out_t, void* stream);
int paradis_pw_gemm_fwd(const float* Wt, const float* WtT /* optional [K,M] copy of Wt, or NULL */,
                        const void* Wsplit /* split image of Wt for `scheme`, NULL for PARADIS_GEMM_EXACT */,
                        int scheme, const uint32_t* x_amax /* amax partials of X: PARADIS_GEMM_F16X2 only */,
                        const float* X, const float* bias, const float* map,
                        const float* m8 /* [cin,N] */, const float* pwT /* [cin,M] */, int cin,
                        /* ^ optional low-rank bias applied on the fly: + sum_c pwT[c,m]*m8[c,n] (GlobalBias
                         *   with projection, reference model/blocks.py:190-196), M % 4 == 0, or NULL/NULL/0 */
                        const float* res, float* Y, float* zpre,
                        int B, int M, int K, int N, int64_t x_bs, int64_t res_bs, int64_t y_bs,
                        int act, void* stream);
/* The same with the gated blend of reference model/paradis.py:239-243 in the epilogue:
 * Y = res + sigmoid(gate[m]) (act(...) - res); gate [M] = alpha_adv, res = the field the advection started from.
 * Bit-identical to paradis_pw_gemm_fwd (res = NULL) followed by paradis_gated_blend_fwd.  (ABI 7) */
int paradis_pw_gemm_fwd_gated(const float* Wt, const float* WtT, const void* Wsplit, int scheme, const uint32_t* x_amax,
                              const float* X, const float* bias, const float* map, const float* m8, const float* pwT,
                              int cin, const float* res, const float* gate, float* Y, float* zpre,
                              int B, int M, int K, int N, int64_t x_bs, int64_t res_bs, int64_t y_bs,
                              int act, void* stream);
/* dX[b][K,N] = (W^T[K,M] * dY[b][M,N]) (* act'(zpre[b][K,N]) if zpre) (+ addend[b][K,N] if addend);
 * WTsplit: split image of W^T (paradis_pw_gemm_split_weights(..., transpose=1)), NULL for PARADIS_GEMM_EXACT */
int paradis_pw_gemm_dgrad(const float* Wt, const void* WTsplit, int scheme,
                          const uint32_t* dy_amax /* amax partials of dY: PARADIS_GEMM_F16X2 only */,
                          const float* dY, const float* zpre,
                          const float* addend, float* dX, int B, int M, int K, int N, int64_t dy_bs,
                          int64_t z_bs, int64_t add_bs, int64_t dx_bs, int act, void* stream);
/* dW[M,K] = sum_b dY[b][M,N] * X[b][K,N]^T ; gbias[M] = sum_{b,n} dY (optional, NULL to skip; fused
 * into the GEMM as row sums of its A operand); workspace >= paradis_pw_gemm_wgrad_ws_bytes;
 * a split scheme is used when N % 16 == 0 and the rows are 16-B aligned (the exact kernels run otherwise) */
size_t paradis_pw_gemm_wgrad_ws_bytes(int B, int M, int K, int N);
/* number of K-range slabs of the split weight-gradient kernel for this shape: 1 or an even number (alternate slabs
 * accumulate with opposite sign; the bf16 MFMA's accumulator-alignment offset cancels in the slab sum).  (ABI 8) */
int paradis_pw_gemm_wgrad_slabs(int B, int M, int K, int N);
int paradis_pw_gemm_wgrad(const float* dY, const float* X, float* dW, float* gbias,
                          int B, int M, int K, int N, int64_t dy_bs, int64_t x_bs, int scheme,
                          const uint32_t* dy_amax, const uint32_t* x_amax /* PARADIS_GEMM_F16X2 only */,
                          void* workspace, void* stream);

/* ---- bf16-STORED tensors of the bf16-mixed mode (ABI 9; PARADIS_GEMM_BF16 only).  Under torch.autocast(bfloat16) the
 * reference's nn.Conv2d calls (model/blocks.py:86,110) produce bf16 TENSORS (config/paradis_settings.yaml:75 ->
 * train.py:56); these entry points take / write such tensors in place of the fp32 words holding bf16 values that the
 * fp32-pointer entry points above use in this scheme: same values bit for bit, half the bytes.  `io16` says which
 * tensors are bf16 (2 bytes per element; all strides stay in ELEMENTS); everything not named is fp32.
 *   fwd16  : PARADIS_IO_X16 = X,  PARADIS_IO_Y16 = Y and zpre (no residual then)
 *   dgrad16: PARADIS_IO_X16 = dY, PARADIS_IO_Y16 = dX, PARADIS_IO_Z16 = zpre
 *   wgrad16: PARADIS_IO_X16 = X,  PARADIS_IO_DY16 = dY (dW, gbias fp32)
 * A bf16 activation operand needs N % 8 == 0 (wgrad: N % 16 == 0) and 16-byte aligned planes: it is staged HBM -> LDS by
 * LDS-DMA and transposed by ds_read_b64_tr_b16 (fwd / dgrad), never touching the vector ALU.  Wsplit / WTsplit: the
 * PARADIS_GEMM_BF16 weight images of paradis_pw_gemm_split_weights. */
#define PARADIS_IO_X16 1
#define PARADIS_IO_Y16 2
#define PARADIS_IO_Z16 4
#define PARADIS_IO_DY16 8
int paradis_pw_gemm_fwd16(const void* Wsplit, const void* X, const float* bias, const float* map, const float* m8,
                          const float* pwT, int cin, const float* res, const float* gate /* or NULL */, void* Y,
                          void* zpre, int B, int M, int K, int N, int64_t x_bs, int64_t res_bs, int64_t y_bs, int act,
                          int io16, void* stream);
int paradis_pw_gemm_dgrad16(const void* WTsplit, const void* dY, const void* zpre, void* dX, int B, int M, int K, int N,
                            int64_t dy_bs, int64_t z_bs, int64_t dx_bs, int act, int io16, void* stream);
int paradis_pw_gemm_wgrad16(const void* dY, const void* X, float* dW, float* gbias, int B, int M, int K, int N,
                            int64_t dy_bs, int64_t x_bs, int io16, void* workspace, void* stream);
/* the producers of a pointwise layer's input writing it as bf16 (round to nearest even: the value the GEMM rounds its
 * operand to): ChannelNorm (model/blocks.py:118-134) and the depthwise stencil of a SepConv (:101-113) */
int paradis_channel_norm_fwd16(const float* x1, const float* x2, const float* w, const float* b, void* y /* bf16 */,
                               float* mean, float* rstd, int B, int C1, int C2, int P, int64_t x1_bs, int64_t x2_bs,
                               float eps, void* stream);
int paradis_dwconv_geo_fwd16(const float* x, const float* w, const float* bias, void* y /* bf16 */, int B, int C, int H,
                             int W, int k, void* stream);
/* ... and their backward reading the cotangent of that bf16 output as a bf16 tensor - the consumer's data gradient, bf16-valued in
 * the reference's autocast backward as well.  Same arguments as paradis_channel_norm_bwd / paradis_dwconv_geo_bwd otherwise; every
 * result fp32 and bit-identical to the fp32 entry point on the widened cotangent.  Supported where the *_ok query returns 1 (the
 * streaming ChannelNorm kernels: every practical shape; the whole-plane stencil kernel: k = 5, W = 64, H <= 32); elsewhere the
 * call fails with rc 1 and the caller widens gy. */
int paradis_channel_norm_bwd16_ok(int B, int C, int P);
int paradis_channel_norm_bwd16(const void* gy /* bf16 */, const float* x1, const float* x2, const float* w, const float* mean,
                               const float* rstd, float* gx1, float* gx2, float* gw, float* gb, int B, int C1, int C2, int P,
                               int64_t x1_bs, int64_t x2_bs, int64_t gx1_bs, int64_t gx2_bs, const float* addend1,
                               int64_t add1_bs, void* workspace, void* stream);
int paradis_dwconv_geo_bwd16_ok(int H, int W, int k);
int paradis_dwconv_geo_bwd16(const void* gy /* bf16 */, const float* x, const float* w, const float* addend, float* gx, float* gw,
                             float* gbias, int B, int C, int H, int W, int k, void* workspace, void* stream);
/* paradis_act_bwd writing gx = gy * act'(x) as a bf16 tensor (to nearest even): d(pre-activation) of a pointwise layer, which both
 * of its gradient GEMMs would round to bf16 on load.  n % 4 == 0, 16-byte aligned gy / x. */
int paradis_act_bwd16(const float* gy, const float* x, void* gx /* bf16 */, int64_t n, int act, void* stream);
/* paradis_bias_grads on a bf16-stored dz (P % 8 == 0, 16-byte aligned rows); outputs fp32 */
int paradis_bias_grads16(const void* dz, float* gmap, float* gbias, int B, int C, int P, int64_t dz_bs, void* stream);

/* ---- a8: ChannelNorm (reference model/blocks.py:118-134): per-pixel, unbiased variance.
 * x may be the virtual concatenation [x1 (C1 ch, batch stride x1_bs) ; x2 (C2 ch)] (paradis.py:249). */
int paradis_channel_norm_fwd(const float* x1, const float* x2, const float* w, const float* b,
                             float* y, float* mean, float* rstd,
                             int B, int C1, int C2, int P, int64_t x1_bs, int64_t x2_bs,
                             float eps, void* stream);
size_t paradis_channel_norm_bwd_ws_bytes(int B, int C, int P);
/* gx1/gx2 receive the slices of the input gradient (gx2 may be NULL); gw,gb [C].
 * addend1 (optional, [B,C1,P] with batch stride add1_bs) is added to gx1: the gradient that reaches
 * x1 through the residual connection around the block (paradis.py:246,253), so autograd's separate
 * accumulation pass disappears. */
int paradis_channel_norm_bwd(const float* gy, const float* x1, const float* x2, const float* w,
                             const float* mean, const float* rstd, float* gx1, float* gx2,
                             float* gw, float* gb, int B, int C1, int C2, int P,
                             int64_t x1_bs, int64_t x2_bs, int64_t gx1_bs, int64_t gx2_bs,
                             const float* addend1, int64_t add1_bs, void* workspace, void* stream);

/* ---- a9: GlobalBias map (reference model/blocks.py:188-196).
 * m8[Cin,H,W] = sum_r A[c,r] U[r,h] V[r,w];  map[Co,H,W] = Pw[Co,Cin] m8 (or map = m8 if Pw NULL). */
int paradis_global_bias_map_fwd(const float* A, const float* U, const float* V, const float* Pw,
                                float* m8, float* map, int Cin, int Co, int R, int H, int W, void* stream);
size_t paradis_global_bias_map_bwd_ws_bytes(int Cin, int Co, int R, int H, int W);
int paradis_global_bias_map_bwd(const float* gmap, const float* A, const float* U, const float* V,
                                const float* Pw, const float* m8, float* gA, float* gU, float* gV,
                                float* gPw, int Cin, int Co, int R, int H, int W,
                                void* workspace, void* stream);

/* GlobalBias in two stages, used when the projection is fused into the GEMM epilogue */
int paradis_global_bias_m8_fwd(const float* A, const float* U, const float* V, float* m8,
                               int Cin, int R, int H, int W, void* stream);
int paradis_global_bias_m8_bwd(const float* gm8, const float* A, const float* U, const float* V,
                               float* gA, float* gU, float* gV, int Cin, int R, int H, int W,
                               void* workspace, void* stream);
int paradis_global_bias_proj_bwd(const float* gmap, const float* m8, const float* Pw, float* gPw,
                                 float* gm8, int Cin, int Co, int64_t P, void* stream);

/* ---- elementwise / reductions used by the blocks */
int paradis_act_fwd(const float* x, float* y, int64_t n, int act, void* stream);
int paradis_act_bwd(const float* gy, const float* x, float* gx, int64_t n, int act, void* stream);
/* out = h + sigmoid(alpha[c]) * (adv - h)          (reference model/paradis.py:239,243) */
int paradis_gated_blend_fwd(const float* h, const float* adv, const float* alpha, float* out,
                            int B, int C, int P, void* stream);
size_t paradis_gated_blend_bwd_ws_bytes(int B, int C, int P);
int paradis_gated_blend_bwd(const float* gout, const float* h, const float* adv, const float* alpha,
                            float* gh, float* gadv, float* galpha, int B, int C, int P,
                            void* workspace, void* stream);
/* the same gradients from the blended OUTPUT instead of adv (the gated GEMM epilogue never writes adv):
 * galpha = (1 - sigmoid) sum gout (out - h)  (ABI 7) */
int paradis_gated_blend_bwd_out(const float* gout, const float* h, const float* out, const float* alpha,
                                float* gh, float* gadv, float* galpha, int B, int C, int P,
                                void* workspace, void* stream);
/* gmap[C,P] = sum_b dz[b,C,P] (NULL to skip), gbias[C] = sum_{b,p} dz (NULL to skip) */
int paradis_bias_grads(const float* dz, float* gmap, float* gbias, int B, int C, int P,
                       int64_t dz_bs, void* stream);
/* y = a + b (n elements) */
int paradis_add(const float* a, const float* b, float* y, int64_t n, void* stream);
/* out[cols,rows] = in[rows,cols]^T (weights for the LDS-DMA forward GEMM) */
int paradis_transpose(const float* in, float* out, int rows, int cols, void* stream);
/* y[b,i] = x[b,i] + m[i], i < per_sample  (standalone GlobalBias.forward, reference model/blocks.py:196) */
int paradis_add_bcast(const float* x, const float* m, float* y, int64_t per_sample, int B, void* stream);

/* ---- rows f1-f3 (callers around the model in the training step)
 * f1: ParadisLoss forward + gradient (reference utils/loss.py:233-282): loss[0] = mean(wf[c]*wl[h]*l(pred-target)),
 *     grad = d loss / d pred (for upstream gradient 1).  kind: 0 mse, 1 smooth reversed Huber.
 *     wl may be NULL; partial needs paradis_loss_blocks(total) floats. */
int paradis_loss_blocks(int64_t total);
int paradis_loss_fwd_bwd(const float* pred, const float* target, const float* wf, const float* wl,
                         float* loss, float* grad, float* partial, int B, int C, int H, int W,
                         int kind, float delta, void* stream);
/* y = x * scalar[0] (device scalar) */
int paradis_scale(const float* x, const float* scalar, float* y, int64_t n, void* stream);
/* f2: dst[b, 0:per_sample] = src[b, 0:per_sample] with different batch strides (channel-block copy used
 *     to assemble cat([input, forcings, constants]) and the next autoregressive input,
 *     reference trainer.py:534-538, 710-729) */
int paradis_copy_channels(const float* src, int64_t src_bs, float* dst, int64_t dst_bs, int B,
                          int64_t per_sample, void* stream);
/* f3: AdamW update of one tensor, torch.optim.AdamW operation order (reference trainer.py:327-335) */
int paradis_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, void* stream);
/* The same update for a whole parameter group in one launch.  Device tables: ptrs[4][n_tensors]
 * (addresses of p, g, m, v as int64), numel[n_tensors], and one entry per chunk of
 * paradis_adamw_chunk() elements: chunk_tensor[n_chunks], chunk_off[n_chunks].
 * dev_state (optional, NULL = use `step` and `lr`): int32[2] on the device, [0] = step count, [1] = the bits of
 * the fp32 learning rate.  With it the bias corrections are formed on the device, so a HIP graph captured around
 * the training step (harness.GraphedTrainStep) stays valid from replay to replay; paradis_adamw_tick advances the
 * count on the stream (once per optimiser step, before the groups' updates). */
int paradis_adamw_chunk(void);
int paradis_adamw_multi(const int64_t* ptrs, const int64_t* numel, const int* chunk_tensor,
                        const int64_t* chunk_off, int n_tensors, int n_chunks, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, const int* dev_state, void* stream);
int paradis_adamw_tick(int* dev_state, void* stream);

/* ---- f3 (second half): Muon / NorMuon step on T same-shaped weight matrices w_t[rows, cols] (conv
 * weights flattened to [out, in*kh*kw]), the reference's default optimiser for Conv/Linear weights
 * (trainer.py:24-64,337-364; `dion` package, un-vendored and un-pinned: restated from its published
 * algorithm - momentum, Frobenius normalisation, 5 quintic Newton-Schulz iterations on the wide
 * orientation, NorMuon's per-neuron second-moment normalisation, decoupled weight decay).
 * ptrs: DEVICE table [4][table_stride] of addresses (int64) of w, g, m (momentum [rows*cols]) and
 * v (NorMuon per-row state [rows]); lr_adj: the shape-adjusted learning rate. */
size_t paradis_muon_ws_bytes(int T, int rows, int cols);
int paradis_muon_step(const int64_t* ptrs, int table_stride, int T, int rows, int cols, float lr,
                      float lr_adj, float mu, float beta2, float weight_decay, float eps, int nesterov,
                      int normuon, int split /* Newton-Schulz products on the bf16-split GEMM */,
                      void* workspace, void* stream);
/* Plain batched GEMM C_b[M,N] = A_b[M,K] B_b[K,N] (row-major; AT = optional [K,M] transposes of A_b,
 * enabling the LDS-DMA kernel; split_ws = optional nbatch * paradis_pw_gemm_split_bytes(M,K,PARADIS_GEMM_BF16X3) bytes of
 * scratch selecting the bf16-split arithmetic) used by the Newton-Schulz iteration. */
int paradis_bgemm(const float* A, const float* AT, const float* B, float* C, int nbatch, int M, int K, int N,
                  int64_t a_bs, int64_t at_bs, int64_t b_bs, int64_t c_bs, void* split_ws, void* stream);

/* ---- f4: data feed on the device --------------------------------------------------------------
 * Forcings of B series of T consecutive timestamps each, every series as the dataset assembles one
 * sample (reference data/era5_dataset.py:587-621; data/forcings/time_vars.py:6-40;
 * data/forcings/toa_radiation.py:38-199):
 * out[B, T-n+1, H, W, n_vars*n] float32, out[b,s,y,x, v*n+k] = forcing v at time s+k of series b.
 * times_us: int64 microseconds since 1970-01-01 (numpy datetime64[us]); lat/lon in degrees (float64
 * storage; lat_is_f32 != 0 reproduces the float32 arithmetic numpy uses when the latitude array is
 * float32).  var_codes (HOST array): 0 toa_incident_solar_radiation (z-scored with toa_mean/std),
 * 1 sin_time_of_day, 2 cos_time_of_day, 3 sin_year_progress, 4 cos_year_progress. */
size_t paradis_forcings_ws_bytes(int B, int T);
int paradis_forcings(const int64_t* times_us /* [B,T] */, const double* lat_deg, const double* lon_deg,
                     int lat_is_f32, int B, int T, int H, int W, int n_time_inputs, const int* var_codes, int n_vars,
                     double toa_mean, double toa_std, float* out, void* workspace, void* stream);
/* Channels-last feature (de)normalisation in place (reference utils/normalization.py:6-80 as applied by
 * data/era5_dataset.py:547-584): data[rows, C]; kind[c]: 0 none, 1 z-score (p0 mean, p1 std),
 * 2 specific humidity (p0 q_min, p1 q_max, eps_q), 3 precipitation (shift 10, eps 1e-6). */
int paradis_normalize_features(float* data, const int* kind, const float* p0, const float* p1,
                               int64_t rows, int C, float eps_q, int inverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARADIS_HIP_H */
