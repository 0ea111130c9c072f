/* rsuper_hip.h -- C ABI of the MI355X-native (gfx950) kernels behind R-Super's training hot path.
 *
 * The reference (MrGiovanni/R-Super) contains no native code and no FFI: every device op of
 * rsuper_train/train_ddp.py's step is an ATen/cuDNN call made from Python.
 * Each entry point below therefore cites the reference Python call site whose ATen op it replaces
 * (paths relative to rsuper_train/).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless named host_*;
 *   - activations are channels-last: element (n,d,h,w,c) of a view lives at x[(((n*D+d)*H+h)*W+w)*ld + c];
 *     C and ld are multiples of 8, base pointers 16-byte aligned;
 *   - dtype: RSUPER_F32 (parity mode, exact-f32 MFMA) or RSUPER_BF16 (bf16 storage, f32 accumulate);
 *   - logits, masks' companions and all parameter/gradient tensors are f32; masks are uint8 0/1;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous, never own their inputs;
 *   - return value: 0 = RSUPER_OK, otherwise an RSUPER_ERR_* code (nothing was launched on error).
 */
#ifndef RSUPER_HIP_H
#define RSUPER_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RSUPER_F32 0
#define RSUPER_BF16 1
#define RSUPER_OK 0
#define RSUPER_ERR_ARG 1
#define RSUPER_ERR_LAUNCH 2
#define RSUPER_ERR_UNSUPPORTED 3
#define RSUPER_ERR_NO_DEVICE 4

const char* rsuper_version(void);
/* 0 when device 0 is a gfx950 GPU; RSUPER_ERR_NO_DEVICE otherwise.  The host layer refuses to run without it. */
int rsuper_device_check(void);

/* ------------------------------------------------------------------------------------------------
 * 3x3x3 convolution (stride 1, pad 1, no bias) -- nn.Conv3d inside ConvNormAct,
 * model/dim3/conv_layers.py:29-38,46-51; BasicBlock model/dim3/conv_layers.py:86-94.
 * ------------------------------------------------------------------------------------------------ */

/* Elements (of `dtype`) of the fragment-ordered weight buffer for GEMM-K sources (ka, kb) and n_cols columns. */
size_t rsuper_conv3_packed_elems(int dtype, int ka, int kb, int n_cols, int bn);

/* Re-order state_dict weights (Cout, Cin, 3,3,3) f32 into MFMA B-fragment order.
 * mode 0 (forward):  K = forward input channels split as ka|kb (concat sources, model/dim3/unet_utils.py:71),
 *                    columns [0,na) from wa and [na,na+nb) from wb (conv1 + shortcut fused, conv_layers.py:79,84).
 * mode 1 (data grad): K = forward output channels (ka rows of wa, kb rows of wb), columns = forward Cin = na;
 *                    taps flipped.  nb = 0: all na columns; nb = (first column << 16) | columns (first column a multiple of 32):
 *                    only that column range, as a buffer of its own (rsuper_conv3_packed_elems with n_cols = columns) -- the data gradient of a
 *                    two-source block as one launch per source (round 5: 96 columns = 32 + 64 run 11 % faster as two launches). */
int rsuper_conv3_pack_weights(int dtype, int mode, const float* wa, const float* wb, int ka, int kb, int na, int nb,
                              int bn, void* packed, void* stream);

/* Same for n layers in one launch (all convolutions of a UNet forward, or of its backward).  host_desc: n x 6 ints
 * (mode, ka, kb, na, nb, bn); host_wa/host_wb: n device pointers each (wb may be NULL); host_out_elems: element offset
 * of every layer's fragment buffer inside `packed`, laid out back to back (each rsuper_conv3_packed_elems long). */
int rsuper_conv3_pack_weights_batch(int dtype, int n, const int* host_desc, const float* const* host_wa, const float* const* host_wb,
                                    const size_t* host_out_elems, void* packed, void* stream);

/* Number of 4x4x16 output tiles per sample == rows of the per-block partial-sum buffer. */
int rsuper_conv3_tiles(int D, int H, int W);

/* Kernel variant of rsuper_conv3_igemm: 0 = classic one-tile-per-block kernel (always used for f32), 1 = wave-specialised
 * producer/consumer persistent kernel (bf16), 2 = per-launch choice between the two (round-1 default), 3 (default) = as 2,
 * with the weight-stationary 8-wave kernel (weights in registers, activation fragments re-used across taps, epilogue
 * straight from the accumulators) for 32-column launches over a single 32-channel K chunk, 4 = weight-stationary kernel
 * for every bf16 32-column launch, 6 / 7 = the volume-fitted
 * K-split kernel (rsuper_conv3_box_bn) forced for every bf16 64-column launch (6: box shape per volume, 7: the 4x4x4 box).
 * v < 0 queries.  Returns the variant in effect. */
int rsuper_conv3_variant(int v);
/* bf16 weight gradients (rsuper_conv3_wgrad) run the second-generation kernel (operand re-use across taps, double-buffered tiles) when a block
 * sweeps at least `t` spatial tiles -- default 12, the measured break-even against the round-3 kernel; 0 = every supported launch (tests),
 * >= 2^20 = never.  Both extremes also switch the small-volume kernel off (batches of at most 64 tiles whose depth slabs fit LDS otherwise run
 * conv3d_wgrad_sv.hip), so that the tests can pin each tile-streaming kernel on small volumes.  t < 0 queries.  Returns the threshold in effect. */
int rsuper_conv3_wgrad2_min_tiles(int t);

/* Launches whose volume cannot fill the chip with 4x4x16-voxel tiles (the 24^3 / 12^3 levels of the UNet at batch 2:
 * model/dim3/unet.py:49-58 after three / four poolings) run a volume-fitted kernel under the default variant: boxes of
 * 4x4x8 or 4x4x4 voxels that divide the volume, GEMM rows = flat voxel index of the box, the block's four waves split the
 * reduction (taps x k-steps) and reduce-scatter their partial tiles through LDS before the same fused epilogue.  It works
 * on 64-column blocks: returns 64 when rsuper_conv3_igemm(dtype, ..., n_cols, bn = 64, N, D, H, W) would take that kernel
 * (pack the weights with bn = 64 then), else 0. */
int rsuper_conv3_box_bn(int dtype, int N, int D, int H, int W, int n_cols);

/* The wide full-resolution layers (more than 32 columns, enough 4x8x16-voxel tiles for one persistent block per CU: up4.0 / the 96^3 level at
 * batch 2) run the depth-reuse kernel (conv3d_igemm_kd.hip: an activation fragment feeds the three kd taps, 0.5 / NF LDS fragment reads per MFMA):
 * returns the block width (64 forward; 64 / 96 / 128 data gradient) rsuper_conv3_igemm(dtype, epi, ..., n_cols, bn, N, D, H, W) takes that kernel
 * with -- pack the weights and size `part` with that bn -- else 0.
 * src_flags bit 0: the launch has two sources of which exactly ONE carries (mean, rstd) (mra == NULL xor mrb == NULL): the depth-reuse kernel stages
 * both sources the same way and does not take such a launch -- the query returns 0 and rsuper_conv3_igemm keeps the kernel that bn gets elsewhere.
 * rsuper_conv3_igemm derives the bit from its own arguments; pass the same value to this query and to rsuper_conv3_part_rows. */
int rsuper_conv3_kd_bn(int dtype, int epi, int N, int D, int H, int W, int n_cols, int src_flags);

/* Volumes of at most 6x6x6 voxels (the 6^3 bottleneck level: model/dim3/unet.py:53, four poolings of a 96^3 patch) run one box
 * per sample with the REDUCTION (32-channel chunks) split over blocks: each block writes its raw f32 tile to a workspace and a
 * second kernel adds the splits and applies the epilogue (deterministic: fixed split order).  The caller registers one device
 * buffer of at least rsuper_conv3_workspace_bytes() PER DEVICE (the registration and every launch refer to the calling thread's
 * current HIP device); launches on one stream share it, launches on two streams of one device must not overlap such
 * convolutions.  Without a registered workspace those volumes take the 4x4x4-box shape.  rsuper_conv3_box_bn returns 32 for them. */
int rsuper_conv3_set_workspace(void* ptr, size_t bytes);
size_t rsuper_conv3_workspace_bytes(void);

/* Rows per sample of the `part` buffer rsuper_conv3_igemm(dtype, epi, ..., n_cols, bn, N, D, H, W) writes under the
 * current variant (classic: one row per tile; producer/consumer: one row per (persistent block, consumer wave row)).
 * src_flags: as for rsuper_conv3_kd_bn (bit 0: one normalised + one raw source). */
int rsuper_conv3_part_rows(int dtype, int epi, int N, int D, int H, int W, int n_cols, int bn, int src_flags);

/* Implicit-GEMM convolution.  epi 0: forward  y = conv(prologue(x)) [+ res]; part <- per-tile (sum, sumsq) of y.
 *                             epi 1: data gradient g = conv(dy, flipped w) * [x_hat > 0]; part <- (sum g, sum g*x_n),
 *                                    where x_hat/x_n come from the forward inputs (ex*, with their mean/rstd emr*).
 * mra/mrb: [N][C][2] (mean, rstd) -> fused InstanceNorm(eps=1e-4)+ReLU prologue (conv_layers.py:40-43); NULL = raw.
 * part:    [N][rsuper_conv3_part_rows][n_cols][2] f32 or NULL.
 * Every activation tensor must be smaller than 4 GiB (N*D*H*W*ld*sizeof < 2^32: buffer-addressed staging); larger
 * volumes are refused with RS_ERR_UNSUPPORTED, also by rsuper_conv3_wgrad. */
int rsuper_conv3_igemm(int dtype, int epi,
                       const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* packed, int n_cols, int bn, int N, int D, int H, int W,
                       void* out, int ldo, const void* res, int ldr, float* part,
                       const void* exa, int elda, int eCa, const float* emra,
                       const void* exb, int eldb, int eCb, const float* emrb, void* stream);
/* The same launch with TWO output tensors: columns [0, out_split) go to `out`, columns [out_split, n_cols) to a tensor `out_part` ELEMENTS behind it (allocate
 * both from one buffer), each with row stride ldo.  For the fused [conv1 | shortcut] GEMM of a BasicBlock whose halves are narrower than a 128-byte line (32 bf16
 * channels: up4.0): every consumer of one half -- conv2's staging, its weight gradient, the ReLU mask of its data gradient, the InstanceNorm-backward tail -- then
 * reads whole lines instead of half-used ones.  out_split a multiple of 32; only launches the depth-reuse kernel takes (rsuper_conv3_kd_bn(...) == bn),
 * RS_ERR_UNSUPPORTED otherwise: the caller falls back to the interleaved output. */
int rsuper_conv3_igemm_split_out(int dtype, int epi,
                                 const void* xa, int lda, int Ca, const float* mra,
                                 const void* xb, int ldb, int Cb, const float* mrb,
                                 const void* packed, int n_cols, int bn, int N, int D, int H, int W,
                                 void* out, int ldo, const void* res, int ldr, float* part,
                                 const void* exa, int elda, int eCa, const float* emra,
                                 const void* exb, int eldb, int eCb, const float* emrb, int out_split, long long out_part, void* stream);
/* The STRIDED member: Conv3d(k=3, stride=2, pad=1, bias=False) of down_block(pool=False) (unet_utils.py:38-39; BasicBlock conv1 + shortcut,
 * conv_layers.py:29-38,82-84) and its data gradient, with the minimal MFMA work (the 27 taps split over the 8 parity classes of the input /
 * of the gradient's output; conv3d_igemm_s2.hip).  FD x FH x FW = the full-resolution grid; the half-resolution one is ceil(F / 2).
 *   mode 1 (forward): xa = x (N, FD, FH, FW, lda) with statistics mra; out (N, OD, OH, OW, ldo) = conv(relu(norm(x))); part: N x
 *                     rsuper_conv3_s2_part_rows(dtype, 1, ...) rows of column sums / sums of squares for the next InstanceNorm; xb unused.
 *   mode 2 (dgrad)  : xa | xb = dy sources on the half grid (no statistics); out (N, FD, FH, FW, ldo) = the gradient w.r.t. relu(norm(x)),
 *                     masked by [x_n > 0] of the forward input exa (N, FD, FH, FW, elda) with statistics emra; part: N x
 *                     rsuper_conv3_s2_part_rows(dtype, 2, ...) rows of the InstanceNorm-backward sums.  packed = rsuper_conv3_pack_weights(mode - 1, ..., bn = 64). */
int rsuper_conv3_s2_part_rows(int dtype, int mode, int Ca, int Cb, int n_cols, int N, int FD, int FH, int FW);
int rsuper_conv3_igemm_s2(int dtype, int mode, const void* xa, int lda, int Ca, const float* mra,
                          const void* xb, int ldb, int Cb, const void* packed, int n_cols, int N, int FD, int FH, int FW,
                          void* out, int ldo, float* part, const void* exa, int elda, const float* emra, void* stream);

/* Weight gradient dW[co][ci][tap] += sum_v dy[v][co] * x_hat[v+tap][ci]   (autograd of the same nn.Conv3d under
 * loss.backward(), train_ddp.py:349).  dy rows [0,Ya) accumulate into dwa (Ya, Ca+Cb, 27), rows [Ya,Ya+Yb) into dwb.
 * dwa/dwb are overwritten.  workspace: splits * 27 * (Ya+Yb) * (Ca+Cb) floats (per-split partial slabs, summed by a
 * reduce kernel -- deterministic, no atomics).  use_tr selects ds_read_b64_tr_b16 operand fetch (bf16). */
/* Number of voxel-tile splits (= partial slabs in `workspace`) rsuper_conv3_wgrad should be called with for this shape:
 * fills the chip with resident blocks for the kernel configuration the launch will pick.  Returns <= 0 on bad arguments. */
int rsuper_conv3_wgrad_splits(int dtype, int Ca, int Cb, int Ya, int Yb, int N, int D, int H, int W);

int rsuper_conv3_wgrad(int dtype, int use_tr,
                       const void* xa, int lda, int Ca, const float* mra,
                       const void* xb, int ldb, int Cb, const float* mrb,
                       const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                       float* dwa, float* dwb, float* workspace, int N, int D, int H, int W, int splits, void* stream);

/* The same in two steps, so that the slab reduction (a small bandwidth-bound kernel nothing on the data-gradient chain waits for)
 * can run on another stream: _partial writes the per-split slabs only, _reduce sums them into dwa / dwb (Cin = Ca + Cb). */
int rsuper_conv3_wgrad_partial(int dtype, int use_tr, const void* xa, int lda, int Ca, const float* mra,
                               const void* xb, int ldb, int Cb, const float* mrb,
                               const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                               float* workspace, int N, int D, int H, int W, int splits, void* stream);
int rsuper_conv3_wgrad_reduce(const float* workspace, int splits, int Cin, int Ya, int Yb, float* dwa, float* dwb, void* stream);
/* _reduce for n weight gradients in ONE launch (arrays of length n: the workspaces written by _partial, their splits, Cin = Ca + Cb, Ya, Yb and the
 * destinations): the per-layer reduction is a ~10 us launch at the dependent-launch floor and the hot loop has 34 of them per step
 * (loss.backward(), train_ddp.py:349 -- every nn.Conv3d weight gradient); nothing but the optimiser / the gradient exchange reads dW. */
int rsuper_conv3_wgrad_reduce_batch(int n, const void* const* workspaces, const int* splits, const int* Cin, const int* Ya, const int* Yb,
                                    void* const* dwa, void* const* dwb, void* stream);
/* The same launch also finalises up to two statistics buffers (rsuper_stats_finalize's arithmetic, bit-identical): in a BasicBlock's backward the data gradient of
 * conv1 leaves its InstanceNorm-backward partial rows and the block's two weight gradients their slabs -- one launch turns both into what the InstanceNorm-backward
 * tail and the optimiser read (autograd of conv_layers.py:86-94; a dependent ~5 us launch less per block and input source).  n in [1, 48], nstats in [0, 2];
 * arrays of length nstats: parts[i] = [sN][snblk][sC][2] f32, souts[i] = [sN][sC][2] f32 (two tables when ssplit > 0), scnt = voxels per sample, smode 0 / 1. */
int rsuper_conv3_wgrad_reduce_batch_stats(int n, const void* const* workspaces, const int* splits, const int* Cin, const int* Ya, const int* Yb,
                                           void* const* dwa, void* const* dwb, int nstats, const void* const* parts, const int* sN, const int* snblk,
                                           const int* sC, const double* scnt, const int* smode, const int* ssplit, void* const* souts, float eps, void* stream);

/* Weight gradient of the STRIDED member (Conv3d(k=3, stride=2, pad=1): BasicBlock(stride=2) conv1 + shortcut of down_block(pool=False),
 * unet_utils.py:18-33, conv_layers.py:60-94):  dW[co][ci][t] = sum_o dy[o][co] * x_hat[2o + t - 1][ci].
 * x (with its mean/rstd) lives on the full-resolution grid (N, FD, FH, FW); dy rows [0,Ya) -> dwa, [Ya,Ya+Yb) -> dwb live on the
 * ((FD+1)/2, (FH+1)/2, (FW+1)/2) grid -- no zero-stuffed operand.  workspace: splits * 27 * (Ya+Yb) * Ca floats, splits from
 * rsuper_conv3_wgrad_s2_splits.  dwa / dwb are overwritten (deterministic slab reduction). */
int rsuper_conv3_wgrad_s2_splits(int dtype, int Ca, int Mtot, int N, int FD, int FH, int FW);
int rsuper_conv3_wgrad_s2(int dtype, const void* xa, int lda, int Ca, const float* mra,
                          const void* ya, int ldya, int Ya, const void* yb, int ldyb, int Yb,
                          float* dwa, float* dwb, float* workspace, int N, int FD, int FH, int FW, int splits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm3d(eps, affine=False) statistics and backward tail -- conv_layers.py:40-42
 * ------------------------------------------------------------------------------------------------ */
/* 1x1x1 convolution / linear layer over the channel axis of f32 channels-last rows, as an MFMA GEMM (bf16 or exact-f32 compute, f32
 * storage): the pointwise members of MedFormer's attention stages -- DepthwiseSeparableConv.pointwise (conv_layers.py:126-157), MBConv
 * expand / project (:197-239), BidirectionAttention's feat_qv / map_qv / out projections (medformer_utils.py:13-99).
 *   mode 0: y[r][n] = sum_k x[r][k] w[n][k] (+ bias[n]),  w = (N, K) as in the state_dict   (F.conv3d(k=1) / F.linear forward)
 *   mode 1: y[r][n] = sum_k x[r][k] w[k][n],              w = (K, N)                        (their data gradient, x := dy)
 * res (nullable, [R][ldr] f32): added after the bias -- the identity shortcut of MBConv (`... + x`, conv_layers.py:239) and of the attention
 * blocks, in the GEMM's epilogue instead of a separate element-wise launch (same fp32 additions in the same order: bit-identical).
 * N, ldx, ldy multiples of 4; R * ldx * 4 < 2^32; packed: device workspace of rsuper_pointwise_packed_bytes(dtype, K, N) bytes. */
size_t rsuper_pointwise_packed_bytes(int dtype, int K, int N);
int rsuper_pointwise(int dtype, int mode, const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr,
                     float* y, int ldy, long R, int K, int N, void* packed, void* stream);
/* The fragments of MANY weights in one launch.  table (device): n + 1 rows of 8 x int64 {weight pointer (f32, row-major rows x cols), byte offset
 * of its fragments in `arena`, rows, cols, mode, ceil(N / 32), k-steps = ceil(K / (dtype == f32 ? 8 : 16)), first item}; an item is one 16-byte
 * fragment slot (64 per (k-step, 32-row tile)), numbered through all entries; row n holds {.., first item = total_items}.  rsuper_pointwise with
 * w == nullptr then takes `packed` = arena + offset as is. */
int rsuper_pointwise_pack_batch(int dtype, const long long* table, int n, long total_items, void* arena, void* stream);
/* Weight and bias gradient of the same layer (autograd of F.linear / Conv3d(k=1) under loss.backward(), train_ddp.py:349):
 *   dw[n][k] = sum_r dy[r][n] x[r][k]  (N, K) f32, overwritten;   db[n] = sum_r dy[r][n] (nullptr: not wanted)
 * The rows are cut into `splits` = rsuper_pointwise_wgrad_splits(R, N, K) slabs whose partial results go through `workspace`
 * (splits * (N*K + N) floats) and are added in slab order (deterministic).  N, K multiples of 4; (R + 1) * ld * 4 < 2^32. */
int rsuper_pointwise_wgrad_splits(long R, int N, int K);
int rsuper_pointwise_wgrad(int dtype, const float* dy, int ldy, const float* x, int ldx, long R, int N, int K, float* workspace, int splits,
                           float* dw, float* db, void* stream);

/* part [N][nblk][C][2] -> out [N][C][2]: mode 0 (mean, rstd = 1/sqrt(var+eps)), mode 1 (sum0/cnt, sum1/cnt).
 * split > 0 writes two contiguous tables instead, [N][split][2] followed by [N][C-split][2] (the column groups of a fused
 * conv1 + shortcut GEMM, or the two sources of a concatenated input), so each can be handed to a kernel as-is. */
int rsuper_stats_finalize(const float* part, int N, int nblk, int C, double cnt, float eps, int mode, int split, float* out, void* stream);
/* dx = rstd * (g - gm0 - x_n * gm1) [+ add1] [+ add2] */
int rsuper_in_bwd_finalize(int dtype, const void* g, int ldg, const void* x, int ldx, const float* mr, const float* gm,
                           const void* add1, int lda1, const void* add2, int lda2, void* out, int ldo,
                           int N, int vox, int C, void* stream);

/* nn.MaxPool3d(2) -- model/dim3/unet_utils.py:35-37.  part: [N][blocks][C][2] partial stats of y (or NULL). */
int rsuper_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                        int N, int D, int H, int W, int C, void* stream);
int rsuper_maxpool2_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx,
                        int N, int D, int H, int W, int C, void* stream);
/* dx = maxpool backward(dy) + add in one pass: x of a down_block is also the skip connection of the matching up_block (model/dim3/unet.py:57-66),
 * `add` is the gradient that arrived through the skip; autograd would store their sum with one more read-read-write launch.  Even D, H, W only
 * (RS_ERR_UNSUPPORTED otherwise).  Bit-identical to the two steps (both round the f32 sum of two bf16 values once). */
int rsuper_maxpool2_bwd_add(int dtype, const void* x, int ldx, const void* dy, int lddy, const void* add, int lda, void* dx, int lddx,
                            int N, int D, int H, int W, int C, void* stream);

/* Stride-(2,2,2) down-sampling of down_block(pool=False) -- model/dim3/unet_utils.py:38-39 (`block(in_ch, out_ch,
 * stride=down_scale)`; conv_layers.py:29-38 with stride 2, padding 1).  The strided convolution is the stride-1 convolution
 * (rsuper_conv3_igemm) evaluated at the even voxels: fwd picks y[od,oh,ow] = x[2od,2oh,2ow] (output ceil(D/2) x ceil(H/2) x
 * ceil(W/2)) and writes the partial stats of y; bwd scatters dy back to the even voxels of a zero-filled dx (D, H, W = the
 * full-resolution dims in both calls). */
int rsuper_subsample2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                          int N, int D, int H, int W, int C, void* stream);
int rsuper_subsample2_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int N, int D, int H, int W, int C, void* stream);

/* F.interpolate(mode='trilinear', align_corners=True) -- model/dim3/unet_utils.py:69 */
int rsuper_upsample_fwd(int dtype, const void* x, int ldx, void* y, int ldy, float* part, int blocks,
                        int N, int ID, int IH, int IW, int OD, int OH, int OW, int C, void* stream);
int rsuper_upsample_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx,
                        int N, int ID, int IH, int IW, int OD, int OH, int OW, int C, void* stream);

/* inconv.conv1 = nn.Conv3d(1, C, 3, padding=1, bias=False) -- model/dim3/unet_utils.py:14.  x: [N][D][H][W] f32. */
int rsuper_stem_fwd(int dtype, const float* x, const float* w, void* y, int ldy, float* part,
                    int N, int D, int H, int W, int C, void* stream);
/* dw (C,1,3,3,3) f32 is overwritten (per-block partial rows + fixed-order reduce: deterministic, no atomics). */
int rsuper_stem_wgrad(int dtype, const float* x, const void* dy, int lddy, float* dw, int N, int D, int H, int W, int C, void* stream);

/* outc = nn.Conv3d(C, K, kernel_size=1) with bias -- model/dim3/unet.py:47.  logits: [N][K][vox] f32 (NCDHW). */
int rsuper_head_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, float* logits, int N, int vox, int C, int K, void* stream);
int rsuper_head_bwd_data(int dtype, const float* dlogits, const float* w, void* dx, int lddx, int N, int vox, int C, int K, void* stream);
/* dw (K,C) and db (K) f32 are overwritten (same reduction scheme as rsuper_stem_wgrad). */
int rsuper_head_bwd_weight(int dtype, const void* x, int ldx, const float* dlogits, float* dw, float* db, int N, int vox, int C, int K, void* stream);
/* Both gradients of the head in one call: dx (bf16 / f32 channels-last) = dlogits^T W, dw / db overwritten.  With bf16 activations, K <= 32 classes and
 * C <= 32 channels this is ONE pass over dlogits (the weight-gradient kernel also multiplies its staged dlogits tile with W on the matrix cores);
 * otherwise the two calls above.  Autograd of outc = nn.Conv3d(C, K, 1) (model/dim3/unet.py:47) under loss.backward() (train_ddp.py:349). */
int rsuper_head_bwd(int dtype, const void* x, int ldx, const float* dlogits, const float* w, void* dx, int lddx, float* dw, float* db,
                    int N, int vox, int C, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss reductions -- training/losses_foundation.py:945-956 (masked BCE), :541-607 (DiceLossMultiClass),
 * :329/:367 (soft volume), :1743-1811 (ball loss BCE with GWRP / background weights).
 * sums[planes][6] f64 (pre-zeroed): S=sum bce*k, A=sum sig*k, B=sum sig*t*k, Cn=sum t*k, F1=sum bce*k*w1, F2=sum bce*k*(1-w2)
 * flags: bit 0 (backward only) dx += instead of dx =; bit 1: `k` holds the dilated UNKNOWN mask and the voxel weight is
 * 1 - k (get_known_voxels, :150-165) -- saves materialising `1 - dilate(unk)`.
 * ------------------------------------------------------------------------------------------------ */
int rsuper_plane_partials_fwd(const float* x, size_t xstride, const uint8_t* t, const uint8_t* k, const float* w1, const uint8_t* w2,
                              double* sums, int flags, int planes, size_t V, void* stream);
int rsuper_plane_partials_bwd(const float* x, size_t xstride, const uint8_t* t, const uint8_t* k, const float* w1, const uint8_t* w2,
                              const float* g, float* dx, int flags, int planes, size_t V, void* stream);
/* The same two passes reading the target planes in the dataset's bit-packed form (`np.packbits(label, axis=0)`,
 * dataset_abdomenatlas_UFO.py:955,970,975 -- class c of sample b is bit 7 - (c & 7) of byte plane b * tP + (c >> 3); tpk = [planes / tC][tP][V], replaces t)
 * and skipping the planes of the dilated unknown map that cannot hold a 1 (kflags[planes] = rsuper_plane_any of the UNdilated map; only with flags bit 1):
 * 1/8 of the label bytes and, for the usual batch, none of the unknown-map bytes cross HBM. */
int rsuper_plane_partials_fwd2(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, double* sums, int flags, int planes, size_t V, void* stream);
int rsuper_plane_partials_bwd2(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, const float* g, float* dx, int flags, int planes, size_t V, void* stream);
/* The forward sums without a float atomic (round 6): every block of the launch stores its six partial sums to pblk[plane][block][6]
 * (blocks = rsuper_plane_partials_blocks(V) per plane; no pre-zeroing) and rsuper_plane_sums_reduce adds them in block order and converts to f32:
 * out[row][6] for `rows` consecutive planes of one or several _fwd3 launches.  Bit-reproducible by construction; one launch less than the atomic form
 * (zero fill + sums + f64 -> f32 conversion). */
int rsuper_plane_partials_blocks(size_t V);
int rsuper_plane_partials_fwd3(const float* x, size_t xstride, const uint8_t* t, const uint8_t* tpk, int tP, int tC, const uint8_t* k, const uint8_t* kflags,
                               const float* w1, const uint8_t* w2, double* pblk, int flags, int planes, size_t V, void* stream);
int rsuper_plane_sums_reduce(const double* pblk, int rows, int nb, float* out, void* stream);
/* Segmentation term from the sums of the (B*C) label planes: loss[0] = scale * ( sum(S*cw)/(B*C*V) + DiceLossMultiClass ),
 * :945-956 with the adaptive-Tversky Dice of :541-607 (alpha_c = clamp(sum_b FP / (sum_b FP + sum_b FN + 1e-5), 0.2, 0.8),
 * dice = TP / (TP + alpha FP + (1-alpha) FN + 1e-5), mean over (b, c) of (1 - dice) * cw).  sums f32 [B*C][6] as produced
 * by rsuper_plane_partials_fwd (cast to f32); cw [B*C] or NULL; dsums [B*C][6] receives d loss / d sums. */
int rsuper_seg_from_sums(const float* sums, const float* cw, int B, int C, size_t V, double scale, float* loss, float* dsums, void* stream);
/* The report terms of calculate_loss from the (R, 6) sums rsuper_plane_partials_fwd produced for them, with their Jacobians, without leaving the
 * device: dice_volume_loss (volume_loss_basic :250-349 with dice_based_volume_loss :352-395) from the first L*B rows (row li*B + b; flags[b][2L] =
 * annotated-tumour flag | segment gate, rvol[b] = reported volume) when use_vol, ball_loss_bce / ball_loss_dice (:1625-1661 for a sample without
 * tumour = L rows, :1793-1811 + DiceLossMultiClass :541-607 for a tumour sample = 1 row) as the mean over the `nplans` plans given as
 * plan[nplans][2] = (kind 0 / 1, first row).  roww[R] = class weight per row.  loss[3] = (ball_loss_bce, ball_loss_dice, dice_volume_loss),
 * jac[3][R][6] = d loss_k / d sums. */
int rsuper_report_from_sums(const float* sums, const float* roww, int R, int B, int L, size_t V, int use_vol, const float* flags, const float* rvol,
                            double tol, double E, int nplans, const int* plan, int apply_dice, int standard_ce, float* loss, float* jac, void* stream);
int rsuper_sigmoid_mask(const float* x, const uint8_t* m, float* out, size_t V, void* stream);

/* Sliding-window inference -- inference/inference3d.py:28-107 (inference_sliding_window) and :8-25
 * (inference_whole_image, assign = 1 with the window covering the volume).  The reference accumulates window
 * probabilities on the host; here acc[BK][D][H][W] (f32, device) receives sigmoid(logits[BK][wd][wh][ww]) at offset
 * (d0, h0, w0): added (assign = 0) or stored (assign = 1).  rsuper_window_normalize divides by the separable window count
 * cd[d] * ch[h] * cw[w] (f32 device vectors of D, H, W entries), the `pred_output /= counter` of :101. */
int rsuper_window_accumulate(const float* logits, float* acc, int BK, int wd, int wh, int ww, int D, int H, int W, int d0, int h0, int w0,
                             int assign, void* stream);
int rsuper_window_normalize(float* acc, const float* cd, const float* ch, const float* cw, long BK, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone InstanceNorm3d(affine=False) [+ ReLU], channels-last f32 [N][vox][C], C % 4 == 0 -- the bare norm layers of
 * MedFormer's attention stages (model/dim3/medformer_utils.py:117-118, :160; conv_layers.py:46-51 for 1x1x1 / depthwise members).
 * stats: part [N][rsuper_cnorm_rows(vox)][C][2]; mode 0 (sum x, sum x^2) -> rsuper_stats_finalize(mode 0) gives mr = (mean, rstd);
 *        mode 1 (sum g, sum g * x_hat) with g = dy * [x_hat > 0 when relu] -> rsuper_stats_finalize(mode 1) gives gm.
 * apply: mode 0 out = x_hat (max(x_hat, 0) when relu); mode 1 out = rstd * (g - gm0 - x_hat * gm1).
 *        mode 2 out = x * mr[1] + mr[0]: a per-(sample, channel) affine map (SEBlock scaling x * s and its gradient, conv_layers.py:159-174).
 * ------------------------------------------------------------------------------------------------ */
int rsuper_cnorm_rows(long vox);
/* one-launch variant for small volumes (statistics + finalize + apply in a block per (sample, 64 channels)):
 * mode 0: out = norm(x) [+ relu], mr_out = (mean, rstd);  mode 1: out = dx from (x, dy, mr). */
int rsuper_cnorm_small(const float* x, const float* dy, const float* mr, float* out, float* mr_out, int N, long vox, int C, int relu, float eps,
                       int mode, void* stream);
int rsuper_cnorm_stats(const float* x, const float* dy, const float* mr, float* part, int N, long vox, int C, int relu, int mode, void* stream);
/* stats -> rsuper_stats_finalize -> apply in one host call (same three launches; part / mr / gm as above): forward writes mr and
 * y = norm(x) [+ relu]; backward writes gm and dx. */
int rsuper_cnorm_forward(const float* x, float* part, float* mr, float* y, int N, long vox, int C, int relu, float eps, void* stream);
int rsuper_cnorm_backward(const float* x, const float* dy, const float* mr, float* part, float* gm, float* dx, int N, long vox, int C, int relu,
                          void* stream);
int rsuper_cnorm_apply(const float* x, const float* dy, const float* mr, const float* gm, float* out, int N, long vox, int C, int relu, int mode,
                       void* stream);

/* SEBlock (model/dim3/conv_layers.py:159-174) on a channels-last f32 tensor [N][vox][C]: y = x * sigmoid(W2 relu(W1 mean(x) + b1) + b2), w1 (r, C),
 * w2 (C, r) as in the state_dict of the two 1x1x1 convolutions.  One host call per direction (channel statistics + finalize, the excitation on the
 * (N, C) vector in one block per sample, the per-(sample, channel) affine apply).  Scratch kept by the caller from forward to backward: ms [N][C][2]
 * (channel means), tab [N][C][2] = (0, s), hbuf [N][r] (hidden activations).  part: rsuper_cnorm_rows(vox) * N * C * 2 floats of workspace.
 * backward: ident = [N][C][2] table of (0, 1); gm, dz1 [N][r], tab2 [N][C][2] are workspaces; dw1 / db1 / dw2 / db2 are overwritten. */
int rsuper_se_forward(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* part, float* ms, float* tab,
                      float* hbuf, float* y, int N, long vox, int C, int r, void* stream);
int rsuper_se_backward(const float* x, const float* dy, const float* ident, const float* w1, const float* w2, const float* ms, const float* tab,
                       const float* hbuf, float* part, float* gm, float* dz1, float* tab2, float* dx, float* dw1, float* db1, float* dw2, float* db2,
                       int N, long vox, int C, int r, void* stream);

/* Re-layout of a logits-like f32 tensor between channels-last [N][vox][C] (C % 4 == 0, C <= 64, the first K channels real) and
 * planar [N][K][vox] -- the `permute(0, 4, 1, 2, 3)` between MedFormer's channels-last deep-supervision head and the (N, K, D, H, W)
 * planes the loss reads (model/dim3/medformer.py:190-194).  to_channels_last = 0: dst planar <- src channels-last;
 * 1: dst channels-last <- src planar, padding channels K..C-1 written as zero (the gradient direction). */
int rsuper_cl_planar(const float* src, float* dst, int N, long vox, int C, int K, int to_channels_last, void* stream);

/* Self-attention core of the SemanticMapFusion transformer (Attention.forward, trans_layers.py:52-84 under medformer_utils.py:239-273):
 *   o = softmax(q k^T * scale) v per (sample, head).  qkv (B, L, 3*heads*dim_head) f32 = q | k | v thirds with head h at columns
 *   [h*dim_head, (h+1)*dim_head) of its third (to_qkv(x).chunk(3, -1) + 'b l (h d) -> b h l d'); o (B, L, heads*dim_head) in the
 *   'b h l d -> b l (h d)' layout; p (B, heads, L, L) = the soft-max rows, written forward and read backward (with o, the forward's
 *   output).  _bwd overwrites d_qkv (same layout as qkv).  One launch per direction, deterministic.  L <= 128, dim_head <= 64
 *   (rsuper_token_attn_supported); other shapes -> RS_ERR_UNSUPPORTED. */
int rsuper_token_attn_supported(int L, int dim_head);
int rsuper_token_attn_fwd(const float* qkv, float* o, float* p, int B, int L, int heads, int dim_head, float scale, void* stream);
int rsuper_token_attn_bwd(const float* qkv, const float* o, const float* p, const float* d_o, float* d_qkv, int B, int L, int heads, int dim_head, float scale,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bidirectional attention between the L voxels of a stage and its T semantic-map tokens, f32 channels-last -- the score / soft-max /
 * mixing core of BidirectionAttention (model/dim3/medformer_utils.py:13-99: the two einsums 'bhid,bhjd->bhij', the soft-max
 * over either axis, 'bhij,bhjd->bhid' / 'bhji,bhjd->bhid' and the head rearranges around them):
 *   S[t,l] = scale * <mq[t], fq[l]>;  f_out[l] = sum_t softmax_t(S[:,l])[t] mv[t];  m_out[t] = sum_l softmax_l(S[t,:])[l] fv[l]
 * fqv [B][L][2*inner], mqv [B][T][2*inner]: q in channels [0, inner), v in [inner, 2*inner), inner = heads * dim_head, channel =
 * dim_head_index * heads + head ('b (dim_head heads) ...'); f_out [B][L][inner], m_out [B][T][inner] in the same channel order.
 * lse [B][heads][T][2] = (max, sum) of the voxel-axis soft-max (saved for backward).
 * Workspaces (n = rsuper_battn_chunks(L, heads)): fwd part B*n*T*inner floats, pms B*n*heads*T*2 floats; bwd part B*n*T*2*inner.
 * Built for (T, dim_head) in {(27, 32), (8, 16)}, heads <= 10 (rsuper_battn_supported); anything else -> RS_ERR_UNSUPPORTED.
 * Deterministic (fixed-order merges, no atomics).
 * ------------------------------------------------------------------------------------------------ */
int rsuper_battn_supported(int T, int dim_head, int heads);
int rsuper_battn_chunks(int L, int heads);
int rsuper_battn_fwd(const float* fqv, const float* mqv, float* f_out, float* m_out, float* lse, float* part, float* pms, int B, int L, int T,
                     int heads, int dim_head, float scale, void* stream);
int rsuper_battn_bwd(const float* fqv, const float* mqv, const float* m_out, const float* lse, const float* d_f_out, const float* d_m_out,
                     float* d_fqv, float* d_mqv, float* part, int B, int L, int T, int heads, int dim_head, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise 3x3x3 convolution (groups = C, stride 1, padding 1, no bias), channels-last f32 [N][D][H][W][C], C % 4 == 0 --
 * DepthwiseSeparableConv.depthwise / MBConv.depthwise of MedFormer (model/dim3/conv_layers.py:126-157, :198-240).
 * w: (C, 1, 3, 3, 3) as in the state_dict.  flip = 1 evaluates the data gradient (x := dy, taps mirrored).
 * wgrad: part = workspace of rsuper_depthwise3_rows(N*D*H*W) * 27 * C floats; dw (C, 1, 3, 3, 3) is overwritten (fixed-order
 * reduction, deterministic).
 * ------------------------------------------------------------------------------------------------ */
int rsuper_depthwise3_rows(long vox);
int rsuper_depthwise3_fwd(const float* x, const float* w, float* y, int N, int D, int H, int W, int C, int flip, void* stream);
int rsuper_depthwise3_wgrad(const float* x, const float* dy, float* part, float* dw, int N, int D, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Binary morphology and selection -- training/losses_foundation.py
 * ------------------------------------------------------------------------------------------------ */
/* dilate_volume :22-46 (iterated ball dilation).  in/out/tmp: nvol volumes of D*H*W bytes; tmp may be NULL when
 * kernel_size <= 7. */
int rsuper_dilate_volume(const uint8_t* in, uint8_t* out, uint8_t* tmp, long nvol, int D, int H, int W, int kernel_size, void* stream);
/* Same, with flags[nvol] (0 = the volume is all zero, e.g. from rsuper_plane_any): flagged-empty volumes are written as zeros without
 * being read -- most label planes carry no unknown / segment voxels.  flags may be NULL (= dense). */
int rsuper_dilate_volume_sparse(const uint8_t* in, uint8_t* out, uint8_t* tmp, const uint8_t* flags, long nvol, int D, int H, int W,
                                int kernel_size, void* stream);
/* isolate_tumor :1423-1445: Gaussian-ball correlation (odd diameter d_odd, std) and first-maximum argmax.
 * best: device u64, pre-zeroed; key = (f32 bits << 32) | (0xFFFFFFFF - linear index). conv_out optional (debug).
 * workspace: rsuper_ball_workspace_floats(D, H, W, d_odd) floats ((d_odd/2 + 1) * D*H*W row sums + one occupancy bit per (z, y)
 * row) -> separable two-stage form (row sums per half width, then a k^2 gather per voxel instead of k^3 taps, over the rows of x
 * that hold a non-zero only: x is zero outside the report's organ segment); NULL -> direct form.  Both are f32; they differ
 * only in summation order. */
long rsuper_ball_workspace_floats(int D, int H, int W, int d_odd);
int rsuper_ball_conv_argmax(const float* x, int D, int H, int W, int d_odd, float std, unsigned long long* best, float* conv_out, float* workspace,
                            void* stream);
/* insert_ball :1336-1385; count (device u32, pre-zeroed) += voxels set. */
int rsuper_insert_ball(uint8_t* out, int D, int H, int W, int cz, int cy, int cx, int d_odd, int half, unsigned int* count, void* stream);
/* The same with the centre decoded on the device from `best` (the argmax key rsuper_ball_conv_argmax wrote): isolate_tumor's argmax -> insert_ball
 * hand-over (losses_foundation.py:1445-1448) without a device->host read.  count: device u32, pre-zeroed, receives the ball's voxel count. */
int rsuper_insert_ball_at(uint8_t* out, int D, int H, int W, const unsigned long long* best, int d_odd, int half, unsigned int* count, void* stream);
/* exact top-k as radix select over non-negative f32 (torch.topk use at :1483-1492); ties -> lower index first. */
int rsuper_radix_hist(const float* x, const uint8_t* m, long V, uint32_t prefix, int shift, unsigned int* hist256, void* stream);
/* need_eq = 0xFFFFFFFF: every element equal to the threshold is selected (parallel); otherwise the first need_eq in index order. */
int rsuper_topk_mark(const float* x, const uint8_t* m, long V, uint32_t thr_bits, unsigned int need_eq, uint8_t* out, void* stream);
/* The same top-k with the radix-select state kept on the device: ONE call, no host round trips (the histogram passes above
 * need four device->host reads per mask).  workspace: 260 u32 on the device.  out[i] = 1 for the k largest of x*m
 * (x >= 0), ties -> lower index first.  1 <= k <= V. */
int rsuper_topk_select(const float* x, const uint8_t* m, long V, unsigned int k, uint8_t* out, unsigned int* workspace, void* stream);
/* nk <= 4 selections over the same (x, m) in one pass sequence (isolate_tumor's t / t_small / t_big masks, :1483-1507): k is a HOST
 * array of nk counts, out holds nk volumes, workspace nk * 260 u32.  clip_to_mask != 0 ANDs every selection with m (voxels outside m
 * rank with value 0 as in the dense top-k but are never marked: the "no tumor_mask value outside the ball" step). */
int rsuper_topk_select_multi(const float* x, const uint8_t* m, long V, const unsigned int* k, int nk, uint8_t* out, unsigned int* workspace,
                             int clip_to_mask, void* stream);
/* GlobalWeightedRankPooling(return_weights, hard_cutoff) :442-535 restricted to the pseudo mask. */
int rsuper_compact(const float* x, const uint8_t* pm, long V, float* vals, uint32_t* idx, unsigned int* n, void* stream);
int rsuper_rank_weights(const float* vals, const uint32_t* idx, unsigned int n, float log2_d, float scale, float* w, void* stream);
/* The same weights from a rank order: w[ids[r]] = 2^(r * log2_d) * scale for r < n (ids = voxel indices sorted by value descending, ties by index
 * ascending -- what rsuper_rank_weights derives by pairwise counting in O(n^2)); identical arithmetic, O(n): the large-tumour path of GWRP. */
int rsuper_rank_assign(const long long* ids, unsigned int n, float log2_d, float scale, float* w, void* stream);
/* flags[p] = any(m[p][:]) for `planes` contiguous byte volumes of V voxels (V % 16 == 0 when planes > 1): the
 * `.sum() > 0` / `.any()` tests of calculate_loss / ball_loss (:1625, :313, :335) at HBM rate. */
int rsuper_plane_any(const uint8_t* m, long planes, long V, uint8_t* flags, void* stream);

/* The per-step input guards of train_epoch (train_ddp.py:311-313: `assert not isnan(img).any()`, `max(img) <= 100`, `min(img) >= -100` -- three
 * device->host synchronisations per step in the reference) as one pass over the f32 tensor x[n] (16-byte aligned) that only ORs device flags:
 * flags[0] any NaN, flags[1] any x > hi, flags[2] any x < lo.  The host mirror (train_ddp.StepGuard) reads the flags asynchronously and raises the
 * reference's assertion no later than one step after the fact. */
int rsuper_guard_range(const float* x, size_t n, float lo, float hi, int* flags, void* stream);
/* The batch consistency checks of calculate_loss (losses_foundation.py:864-869) on device flags: per sample b with m_any[b] != 0 (chosen segment mask not all
 * zero; bytes from rsuper_plane_any), flags[0] |= (u_any[b] == 0) ("unk_voxels should not be all zeros ..."), flags[1] |= (sum_t volumes[b][t] == 0)
 * ("tumor_volumes_report should not be all zeros ...").  volumes: [B][T] f32. */
int rsuper_guard_consistency(const uint8_t* m_any, const uint8_t* u_any, const float* volumes, int B, int T, int* flags, void* stream);
int rsuper_mask_op(uint8_t* a, const uint8_t* b, long V, int op /*0 and, 1 or, 2 andnot*/, void* stream);

/* Bit-packed label ingestion: device-side np.unpackbits(packed, axis=0)[:C] of the label / unk / chosen-
 * segment volumes the dataset stores with np.packbits(axis=0) -- training/dataset/dim3/dataset_abdomenatlas_UFO.py:955,
 * 970,975 (pack), :1031-1034 (unpack).  packed: [B][P = ceil(C/8)][V] bytes (MSB = lowest class), out: [B][C][V] 0/1. */
int rsuper_unpack_bits(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, void* stream);
/* Timing events for SURVEY 8(d)'s per-launch measurement (bench.py's roofline pass: HIP events around every conv MFMA launch, on the launch stream): events
 * created with hipEventDisableSystemFence -- a default event performs a system-scope release (cache write-back + invalidate) when recorded, which is charged to the
 * bracket and leaves the next kernel a cold L2; the HIP header recommends the flag for events that only measure time.  elapsed_ms waits for `b`. */
int rsuper_timer_event_create(void** ev);
int rsuper_timer_event_record(void* ev, void* stream);
int rsuper_timer_event_elapsed_ms(void* a, void* b, float* ms);
int rsuper_timer_event_destroy(void* ev);

/* The report losses read only a few planes of a bit-packed volume ("loss kernels should read the packed u8 directly"; losses_foundation.py:286-297,
 * 1571-1605 index the lesion channels of label / unknown / segment mask): plane (b, c) of out[B][C][V] is written iff flags[b * C + c] != 0 or force[c] != 0
 * (flags: [B * C] device bytes or NULL, force: [C] device bytes or NULL; both NULL = rsuper_unpack_bits); the other planes are left untouched, byte planes
 * without a wanted class are not read. */
int rsuper_unpack_bits_sel(const uint8_t* packed, uint8_t* out, int B, int P, int C, long V, const uint8_t* flags, const uint8_t* force, void* stream);
/* flags[b * C + c] = any(class c of sample b) from the packed bytes (OR over byte plane (b, c >> 3), bit 7 - (c & 7)): rsuper_plane_any on the inflated volume
 * at 1/8 of the bytes.  packed 16-byte aligned, V % 16 == 0.  Use: the kflags of rsuper_plane_partials_*2 / the flags of rsuper_dilate_volume_sparse for the
 * unknown-voxel map (get_known_voxels, losses_foundation.py:150-165), the mask / unknown consistency guard (:864-869). */
int rsuper_plane_any_bits(const uint8_t* packed, int B, int P, int C, long V, uint8_t* flags, void* stream);
int rsuper_zero_where(float* x, const uint8_t* m, long V, void* stream);
int rsuper_count(const uint8_t* m, long V, unsigned int* count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser -- train_ddp.py:352-357, training/utils.py:46-51,154-161.  host_* are HOST arrays of device pointers.
 * ------------------------------------------------------------------------------------------------ */
/* *total_sq = sum of squared elements of the n gradient tensors (f64, deterministic order).  The accumulator need not be zeroed: the first launch
 * assigns it -- and no hipMemsetAsync is issued, so the call is safe to capture in a hipGraph (a captured memset node writes 0xC0 bytes after eager
 * interludes on ROCm 7.2: DESIGN.md 3.4c).  The same holds for rsuper_plane_any's flags and rsuper_maxpool2_bwd's zero fill. */
int rsuper_grad_sqnorm(int n, void* const* host_g, const size_t* host_numel, double* total_sq, void* stream);
int rsuper_clip_scale(int n, void* const* host_g, const size_t* host_numel, float max_norm, const double* total_sq, void* stream);
/* One fused pass: clip (coef from *total_sq, NULL = no clipping) + AdamW + EMA (host_ema NULL = no EMA). */
int rsuper_adamw_ema_step(int n, void* const* host_p, void* const* host_g, void* const* host_m, void* const* host_v,
                          void* const* host_ema, const size_t* host_numel, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float ema_alpha, float max_norm, const double* total_sq, void* stream);
/* Same update with the step-dependent scalars read on the device: dyn = [lr, lr / (1 - beta1^t), sqrt(1 - beta2^t), ema_alpha] (f32).
 * A training step captured in a hipGraph replays this launch unchanged while the host refreshes the four floats before each replay. */
int rsuper_adamw_ema_step_dyn(int n, void* const* host_p, void* const* host_g, void* const* host_m, void* const* host_v,
                              void* const* host_ema, const size_t* host_numel, float beta1, float beta2, float eps, float weight_decay,
                              float max_norm, const double* total_sq, const float* dyn, void* stream);

#ifdef __cplusplus
}
#endif
#endif
