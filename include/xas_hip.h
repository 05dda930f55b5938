/*
 * xas_hip.h - C ABI of libxas_hip.so: the MI355X (gfx950) kernels behind the
 * X-as-Supervision training step.
 *
 * The reference (Charrrrrlie/X-as-Supervision) is pure Python/PyTorch and has no FFI of
 * its own; the boundary a maintainer binds is "one call per fused op", replacing the
 * chains of ATen/cuDNN kernels listed beside each entry (reference file:line).  The
 * Python host side (x-as-supervision_amd/xas_amd) binds these with ctypes and wraps
 * them in torch.autograd.Function; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless typed otherwise; buffers are
 *     owned by the caller; the library allocates nothing and never synchronises;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it;
 *   - activations are NHWC ("channels last"): x[n][h][w][c];
 *   - return value 0 = enqueued; != 0 = rejected on the host before any launch
 *     (bad shape / unsupported size), message via xas_last_error().
 */
#ifndef XAS_HIP_H
#define XAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* xas_last_error(void);
/* Kernel-variant selectors kept for coverage tests (tests/test_gpu_nn.py::test_conv_kernel_variants): 0 = the shipped
 * configuration; results agree to accumulation-order noise under every flag.  The first group only concerns the exact-fp32
 * kernels (XAS_PREC_F32).
 *   32      plain K-loop in fwd / dgrad instead of the pipelined one      524288  same for the weight gradient
 *   64      global-load fwd / dgrad kernels (the >= 2 GiB fallback)       128     same for the weight gradient
 *   8192    plain (not XCD-grouped) weight-gradient block order
 *   32768 / 65536 / 98304  column-reduce slab target 512 / 128 / 64       262144  86-VGPR build of the backward column sums
 * and two that concern the bf16-split kernels (tests/test_gpu_tap_kernels.py compares both settings):
 *   4194304 (bit 22)  no tap re-use kernels: stride-1 3x3 layers on the implicit-GEMM kernels (forward, data and weight gradient)
 *   8388608 (bit 23)  no 64 x 256 tiles for layers whose output channels are a multiple of 256
 *   16777216 (bit 24) the general weight-gradient kernel for the 7x7 stem instead of stem_wgrad_kernel
 *   33554432 (bit 25) the exact-fp32 stem forward kernel in the f16x3 mode too */
int xas_set_tuning(int flags);
/* Arithmetic of the MFMA convolutions (forward, data gradient, weight gradient).  All modes keep fp32 activations, fp32
 * master weights and fp32 accumulation; they differ in how a product of two fp32 operands is formed:
 *   XAS_PREC_F16X3 (default)  every fp32 operand is split into two fp16 pieces x = h1 + h2 + e, |e| <= 2^-22 |x| (round to
 *                   nearest twice), after an exact power-of-two scaling that keeps both pieces inside fp16's normal
 *                   range: weights as 2^10 w (|w| < 64; anything larger raises xas_f16_weight_overflow), and EVERY tensor
 *                   operand - activation or gradient - at the scale that puts its maximum in [2^14, 2^15): the maximum comes
 *                   with the call (xas_conv_shape.grad_amax / x_amax), recorded by the kernel that wrote the tensor
 *                   (xas_bn_apply_amax, xas_bn_bwd_apply_amax, xas_head_softargmax_bwd_amax) or by xas_abs_max.  There is
 *                   no fixed activation scale, hence no range activations must stay in (r04).  Three partial products
 *                   h1 g1 + h1 g2 + h2 g1, each exact in fp32, accumulated by v_mfma_f32_32x32x16_f16: 3 instead of 6 matrix
 *                   instructions per K = 16.  Measured distance to a float64 convolution: the same as the exact-fp32 MFMA
 *                   path's (its own accumulation error dominates: 5e-7 relative at K = 576).  A launch that comes WITHOUT
 *                   the maxima of its tensor operands runs as bf16x6 (range-free by construction);
 *   XAS_PREC_BF16X6  every fp32 operand is split exactly into three bf16 pieces and six exact partial products
 *                   are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: per-product error below one fp32 rounding
 *                   (1.09e-7 relative against float64 at K = 64, exact-fp32 MFMA 1.06e-7), 2.67x the fp32-MFMA math rate; no
 *                   assumption on operand ranges;
 *   XAS_PREC_F32    v_mfma_f32_32x32x2_f32, bit for bit a k-ordered fmaf chain;
 *   XAS_PREC_BF16   operands rounded to bf16 once (NOT fp32 accurate; a variant that is reported separately).
 * xas_set_precision sets the process default; a call overrides it with xas_conv_shape.mode = 1 + XAS_PREC_* (0 = default).
 * In the split modes forward / data gradient take PRE-SPLIT weights: see xas_conv_weight_planes / xas_split_weight. */
enum { XAS_PREC_F32 = 0, XAS_PREC_BF16 = 1, XAS_PREC_BF16X6 = 2, XAS_PREC_F16X3 = 3 };
/* A RECORDED MAXIMUM ("amax slot": xas_conv_shape.grad_amax / x_amax, the amax_out arguments) is an array of
 * XAS_AMAX_SLOT_FLOATS floats in device memory, 16-byte aligned: XAS_AMAX_SUB sub-maxima XAS_AMAX_STRIDE floats (128 bytes)
 * apart; the value is the maximum over the sub-maxima (the other floats are not touched).  Producers merge with one atomic
 * per block into the sub-maximum the block index selects - one address per tensor serialised 16 384 atomics per launch, 190 us.
 * ZERO the whole array before the first producer merges into it.  To hand over a maximum computed elsewhere, write it to
 * element 0 of a zeroed array. */
#define XAS_AMAX_SUB 32
#define XAS_AMAX_STRIDE 32
#define XAS_AMAX_SLOT_FLOATS (XAS_AMAX_SUB * XAS_AMAX_STRIDE)
enum { XAS_GRAD_IS_X = 0x100 };      /* flag of xas_conv_shape.mode */
int xas_set_precision(int mode);
int xas_get_precision(void);
int xas_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Soft-argmax ("integral") head.
 * Replaces keypoint_detector_integral_multi.py:69-88 (softmax, six marginal sums, peak
 * pick, topk, two avg_pool1d, two gathers) and keypoint_detector_integral.py:48-63.
 *
 * logits  [B][H][W][K*D] (NHWC storage of the reference's [B, K*D, H, W]); D==H==W.
 * num_hypo >= 1 with neighbor > 0 : multi-hypothesis head; num_hypo == 1 and
 * neighbor == 0 : single-hypothesis head (plain expectation along depth).
 * kps      [B][num_hypo][K][3]   normalised to [-1,1)
 * z_idx    [B][K][num_hypo] int64 (depth-peak bins, 1..D-2; ties: lower index first)
 * depth_prob_map [groups][K][D]  (depth marginal of the FIRST sample of each of `groups` equal sub-batches:
 *                                 sample 0 for groups = 1, multi.py:45; one per camera in a camera-batched pass)
 * stats    [B][K][XAS_HEAD_STATS] saved for backward (lse, X, Y, Z_h, S_h ...)
 * partial  workspace, xas_head_workspace_floats(B,K,D) floats
 * ---------------------------------------------------------------------------------- */
#define XAS_HEAD_STATS 16
size_t xas_head_workspace_floats(int B, int K, int D);
int xas_head_softargmax_fwd(const float* logits, int B, int K, int D, int num_hypo, int neighbor,
                            float* kps, int64_t* z_idx, float* depth_prob_map, int groups, float* stats,
                            float* partial, void* stream);
/* grad_logits [B][H][W][K*D] = d loss / d logits given grad_kps [B][num_hypo][K][3].
 * coef: workspace of B*K*(4+D) floats. */
/* Second pass of xas_head_softargmax_fwd over partial records produced elsewhere (xas_conv_fwd_head): same outputs. */
int xas_head_softargmax_from_partials(const float* partial, int B, int K, int D, int nchunk, int num_hypo, int neighbor,
                                      float* kps, int64_t* z_idx, float* depth_prob_map, int groups, float* stats,
                                      void* stream);

int xas_head_softargmax_bwd(const float* logits, const float* stats, const int64_t* z_idx,
                            const float* grad_kps, int B, int K, int D, int num_hypo, int neighbor,
                            float* grad_logits, float* coef, void* stream);
/* Same, and max |grad_logits| is merged into *amax_out (device float, zeroed by the caller; NULL: none): the scale of the
 * gradient operand of the final 1x1 convolution's f16x3 gradient launches (xas_conv_shape.grad_amax). */
int xas_head_softargmax_bwd_amax(const float* logits, const float* stats, const int64_t* z_idx,
                                 const float* grad_kps, int B, int K, int D, int num_hypo, int neighbor,
                                 float* grad_logits, float* coef, float* amax_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Patch -> world geometry, all hypotheses in one launch.
 * Replaces modules/util.py:128-152 -> :61-95 (clone/index_put chains + two batched
 * torch.linalg.inv) called per camera per hypothesis at modules/model.py:72-84.
 * kps [B][Hy][K][3]; trans_image [B][2][3]; k_mat [B][3][3]; pelvis [B][3];
 * rot_world [B][3][3]; trans_world [B][3]; world [B][Hy][K][3].
 * flags: bit0 is_norm, bit1 mono, bit2 patch (reference defaults: is_norm|patch).
 * ---------------------------------------------------------------------------------- */
#define XAS_GEO_NORM 1
#define XAS_GEO_MONO 2
#define XAS_GEO_PATCH 4
#define XAS_GEO_IMAGE 8   /* forward only: stop after patch -> image (u px, v px, depth mm), util.py:61-83 */
int xas_patch_to_world_fwd(const float* kps, const float* trans_image, const float* k_mat,
                           const float* pelvis, const float* rot_world, const float* trans_world,
                           int B, int Hy, int K, float image_size, float rect_width, int flags,
                           float* world, void* stream);
int xas_patch_to_world_bwd(const float* kps, const float* grad_world, const float* trans_image,
                           const float* k_mat, const float* pelvis, const float* rot_world,
                           const float* trans_world, int B, int Hy, int K, float image_size,
                           float rect_width, int flags, float* grad_kps, void* stream);

/* ------------------------------------------------------------------------------------
 * Line-mask renderer fused with the max over lines.
 * Replaces modules/util.py:21-59 (about 25 ATen kernels, 419 MB intermediates) plus
 * torch.max(dim=1) at modules/model.py:94,96.
 * kps: joints, (x,y) of joint j of sample b at kps[b*kp_stride_b + j*kp_stride_j + {0,1}].
 * parents/children: L int32 each (device).  fine_mask: bit l set -> exponent x2
 * (modules/util.py:50-53, applied by the host when L >= 21).
 * mask [B][1][S][S].   partial: workspace B*nblk*K*2 floats, nblk = xas_lines_nblk(S).
 * ---------------------------------------------------------------------------------- */
int xas_lines_nblk(int S);
int xas_draw_lines_max_fwd(const float* kps, long kp_stride_b, long kp_stride_j, int B, int K,
                           const int* parents, const int* children, int L, unsigned fine_mask,
                           float body_width, int S, float* mask, void* stream);
int xas_draw_lines_max_bwd(const float* kps, long kp_stride_b, long kp_stride_j, int B, int K,
                           const int* parents, const int* children, int L, unsigned fine_mask,
                           float body_width, int S, const float* grad_mask, float* partial,
                           float* grad_kps_xy /* [B][K][2] */, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution family: MFMA implicit GEMM on NHWC fp32 activations (fp32-accurate f16x3 by default - bf16x6 for launches
 * without operand maxima -, exact fp32 MFMA on request: XAS_PREC_*).
 * Replaces the cuDNN calls behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear in
 * integral_base_modules/resnet.py:16-47, deconv_head.py:24-35, physique_network.py:15-50,
 * discriminator.py:8-21 and torchvision's Bottleneck.
 *
 * x  [N][Hi][Wi][Cin]      w  packed [Cout][R][S][Cin] (see xas_pack_weight)
 * y  [N][Ho][Wo][Cout]     Ho = (Hi + 2*pad - R)/stride + 1
 * bias (Cout) may be NULL.  Shapes outside the MFMA tiles (Cin % 32 != 0 or Cout < 16:
 * the 3-channel stem, the 1-channel mask convs) run a direct VALU kernel.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int N, Hi, Wi, Cin;
  int Cout, R, S;
  int stride, pad;
  int Ho, Wo;
  int mode;          /* low byte: 0 = process default precision (xas_set_precision), otherwise 1 + XAS_PREC_* for this call;
                      * | XAS_GRAD_IS_X: in a weight-gradient call grad_amax describes the `x` argument and x_amax the `dy`
                      *   argument (the weight gradient of a ConvTranspose2d, whose gradient tensor is passed as `x`) */
  const float* grad_amax;   /* NULL, or a DEVICE pointer to max |v| over the tensor operand of the call, complete on the call's
                              * stream: the input x of a forward-type launch (xas_conv_fwd*), dy of a data-gradient-type launch
                              * (xas_conv_dgrad*; for a ConvTranspose2d forward that is its input activation), dy of a weight
                              * gradient.  An upper bound is as good as the maximum.  XAS_PREC_F16X3 splits that tensor into two fp16
                              * pieces at the power-of-two scale that puts the maximum just below 2^15; without it the launch runs
                              * as bf16x6.  Producers: xas_bn_apply_amax, xas_bn_bwd_apply_amax, xas_head_softargmax_bwd_amax,
                              * xas_abs_max. */
  const float* x_amax;      /* weight gradient only: the same for its `x` argument (both maxima -> f16x3, else bf16x6) */
} xas_conv_shape;

/* Which weight buffer xas_conv_fwd* (pass 0) / xas_conv_dgrad* (pass 1) expect for this shape in its precision mode:
 * 0 = fp32 packed weights (xas_pack_weight); 1 or 3 = that many bf16 planes, 2 = two fp16 planes of 2^10 w: xas_split_weight of the packed weights
 * [rows][K] with K = R*S*Cin (pass 0, rows = Cout) or R*S*Cout (pass 1, rows = Cin).  The split layout is the MFMA operand
 * image [rows / 32][K / 16][planes][64 lanes][8 bf16] (rows zero-padded to a multiple of 32): the kernels load it straight
 * into operand registers.  Shapes outside the MFMA tiles (stem, one-channel layers) always take fp32 weights.  Built once
 * per optimizer step and weight. */
int xas_conv_weight_planes(const xas_conv_shape* s, int pass);
/* Kernel family a pass of this shape runs on (0 forward-type, 1 data-gradient-type, 2 weight gradient): 0 no MFMA (direct /
 * one-channel kernels), 1 exact-fp32 MFMA, 2 bf16 MFMA, 3 bf16x6 MFMA, 4 f16x3 MFMA.  For measurement (bench.py prices a launch against
 * the peak of the pipe it uses). */
int xas_conv_kernel_class(const xas_conv_shape* s, int pass);
/* f16x3: 1 if a weight of magnitude >= 64 (or a NaN) has gone through xas_split_weight / xas_prepare_weights since the flag
 * was last cleared - such a weight does not fit the 2^10 w scaling of the fp16 pieces and the results of its layer are
 * inf / NaN; 0 otherwise; -1 on a HIP error.  SYNCHRONISES the device (call it rarely: engine.TrainStep every 64 steps).
 * reset != 0 clears the flag.  Remedy: XAS_PREC_BF16X6. */
int xas_f16_weight_overflow(int reset);
/* The same flag WITHOUT a synchronisation: writes it (0 / 1) to the device word *device_out on `stream`; the flag is not
 * cleared.  The caller copies the word to pinned host memory asynchronously and reads it once that copy has completed:
 * engine.TrainStep looks at it one step late, every step, so that a weight leaving the range is reported within a step or two
 * instead of at the next 64-step poll. */
int xas_f16_weight_overflow_peek(unsigned* device_out, void* stream);
size_t xas_split_weight_bytes(long rows, long K, int pieces);
int xas_split_weight(const float* w_packed, void* w_split, long rows, long K, int pieces, void* stream);

/* Pack + split of MANY layers in one launch (the head of every optimisation step: the reference hands cuDNN the fp32
 * parameters, modules/integral_base_modules/resnet.py:16-47; this library wants fragment-ordered bf16 planes of them, rebuilt
 * after every optimizer step).  descs: device array of n entries of 12 int64 each:
 *   { src (OIHW fp32, device), dst (device, xas_split_weight_bytes(rows, K, planes) bytes), Cout, Cin, R, S,
 *     transposed (0: rows = Cout, K = (r, s, ci); 1: rows = Cin, K = (r, s, co), as xas_pack_weight),
 *     planes (3, 2 or 1), rows, K, first block of the entry (prefix sum of ceil(ceil(rows / 32) * (K / 16) * 64 / 256)), 0 }
 * blocks: total number of blocks.  Results are bit-identical to xas_pack_weight followed by xas_split_weight. */
int xas_prepare_weights(const void* descs, int n, long blocks, void* stream);

int xas_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                 const xas_conv_shape* s, void* stream);

/* Final 1x1 convolution of the detector with the soft-argmax head's first pass in its epilogue (forward half):
 * replaces modules/integral_base_modules/deconv_head.py:34-35 followed by the softmax / marginal reductions of
 * modules/keypoint_detector_integral_multi.py:36-62,70-74.  y (the logits, kept for the backward) is written as by xas_conv_fwd;
 * head_partial receives [N][chunks][K][3 + D] floats, chunks = xas_conv_fwd_head_chunks(s, K, D) records of 64 pixels per image
 * (0: the shape is not taken - call xas_conv_fwd and xas_head_softargmax_fwd).  Finish with xas_head_softargmax_from_partials. */
int xas_conv_fwd_head_chunks(const xas_conv_shape* s, int K, int D);
int xas_conv_fwd_head(const float* x, const float* w_packed, const float* bias, float* y, const xas_conv_shape* s,
                      int K, int D, float* head_partial, void* stream);
/* Bias-free convolution followed by training-mode batch norm (resnet.py:17-18 conv1/bn1 and every torchvision Bottleneck
 * conv/bn pair, resnet.py:2): y = conv(x, w) AND the per-group batch statistics of y in one call - the conv epilogue
 * emits per-tile column sums, so y is not read again for its statistics (falls back to xas_conv_fwd + xas_bn_stats when
 * the tile grid does not line up with the groups).  pivot [Cout] or NULL: a value near the channel means (the running
 * mean) that the sums are taken around.  mean / var_biased / out_stride / count_out / running_* / momentum: as
 * xas_bn_stats.  workspace: xas_conv_fwd_bnstats_workspace_floats(s, groups). */
size_t xas_conv_fwd_bnstats_workspace_floats(const xas_conv_shape* s, int groups);
int xas_conv_fwd_bnstats(const float* x, const float* w_packed, float* y, const xas_conv_shape* s, int groups,
                         const float* pivot, float* mean, float* var_biased, long out_stride, float* count_out,
                         float* workspace, float* running_mean, float* running_var, float momentum, void* stream);
/* Backward of  h = relu(batch_norm(xb)) -> conv(h)  (torchvision Bottleneck conv/bn/relu chains, resnet.py:2) in one call:
 * data gradient of the convolution `s` (dy: gradient of its output, w_packed_t as xas_conv_dgrad), ReLU mask re-derived
 * from xb, and the rank-local batch-norm backward (mean / var_biased [groups][Cin], count = rows per group).
 * Outputs: dx = gradient wrt xb; sums [groups][2][Cin] (sum dz | sum dz * xhat); dbeta_acc / dgamma_acc (both or neither
 * NULL) += local parameter gradients; dz: scratch of xb's size.  workspace:
 * xas_conv_dgrad_bn_bwd_workspace_floats(s, groups).  The reductions ride in the data gradient's epilogue when its tile
 * grid lines up with the groups (stride 1), otherwise the call is xas_conv_dgrad + xas_bn_bwd_reduce + xas_bn_bwd_apply.
 * Operand maxima in `s` are IGNORED: w_packed_t is always the format xas_conv_weight_planes(s, 1) names for the shape with
 * grad_amax == NULL (three bf16 planes in both split modes - the fused epilogue exists for that format only). */
size_t xas_conv_dgrad_bn_bwd_workspace_floats(const xas_conv_shape* s, int groups);
int xas_conv_dgrad_bn_bwd(const float* dy, const float* w_packed_t, const xas_conv_shape* s, const float* xb,
                          const float* mean, const float* var_biased, const float* gamma, const float* beta, float eps,
                          int groups, double count, float* dz, float* dx, float* sums, float* workspace, float* dbeta_acc,
                          float* dgamma_acc, void* stream);
/* dx = conv_transpose(dy, w): also the FORWARD of nn.ConvTranspose2d (deconv_head.py:27-29)
 * with roles swapped.  w_packed_t: [Cin][R][S][Cout] (xas_pack_weight transposed=1). */
int xas_conv_dgrad(const float* dy, const float* w_packed_t, float* dx,
                   const xas_conv_shape* s, void* stream);
/* Same, but dx += result: the residual-branch gradient is already in dx (one read-modify-write instead of a
 * separate gradient-accumulation kernel per bottleneck).  MFMA path only (Cout % 32 == 0, Cin % 4 == 0, Cin >= 16). */
int xas_conv_dgrad_acc(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* shape, void* stream);
/* dx = dgrad(dy, W) + relu'(mask) * dprev: the block-input gradient of a bottleneck without a projection, with the skip
 * gradient formed in the epilogue from the block-OUTPUT gradient dprev [like dx] and the sign bytes xas_bn_apply wrote for
 * the block output (one byte per float4 of dx): the skip gradient tensor is never materialised.  Same shape limits as
 * xas_conv_dgrad_acc; dx is written, not read. */
int xas_conv_dgrad_acc_masked(const float* dy, const float* w_packed_t, float* dx, const xas_conv_shape* s,
                              const float* dprev, const uint8_t* mask, void* stream);
/* dw_packed [Cout][R][S][Cin] (+)= sum_n,ho,wo dy * x ; workspace for split-K partials. */
size_t xas_conv_wgrad_workspace_floats(const xas_conv_shape* s);
int xas_conv_wgrad(const float* x, const float* dy, float* dw_packed, float* workspace,
                   const xas_conv_shape* s, void* stream);
/* same, but the split slabs are summed straight into the parameter's OIHW layout [Cout][Cin][R][S] */
int xas_conv_wgrad_oihw(const float* x, const float* dy, float* dw_oihw, float* workspace,
                        const xas_conv_shape* s, void* stream);
/* same, ACCUMULATING: dw_oihw += gradient (used to add straight into the parameter's .grad arena) */
int xas_conv_wgrad_acc(const float* x, const float* dy, float* dw_oihw, float* workspace,
                       const xas_conv_shape* s, void* stream);
/* OIHW [Cout][Cin][R][S] <-> packed.  transposed=0: [Cout][R][S][Cin];
 * transposed=1: [Cin][R][S][Cout].  unpack adds nothing: it overwrites dst. */
int xas_pack_weight(const float* oihw, float* packed, int Cout, int Cin, int R, int S,
                    int transposed, void* stream);
int xas_unpack_weight(const float* packed, float* oihw, int Cout, int Cin, int R, int S,
                      int transposed, void* stream);

/* ------------------------------------------------------------------------------------
 * Batch normalisation (training mode), activation, residual; NHWC, M = N*H*W rows.
 * Replaces ATen batch_norm_stats / batch_norm_elemt / batch_norm_backward_{reduce,elemt}
 * behind nn.BatchNorm2d and nn.SyncBatchNorm (resnet.py:18,40, deconv_head.py:30,
 * physique_network.py:18,25,33).
 * GROUPS.  The M rows may be `groups` independent batches of M / groups consecutive rows, each normalised with
 * its own statistics: the camera-batched step sends the images of all cameras through the detector as one
 * tensor while every camera keeps its own batch statistics and its own running-statistic update, exactly as
 * the reference's per-camera calls (model.py:64,147,231).  Statistics are [groups][C]; groups = 1 is the plain
 * layer.  Running statistics receive one update per group, in group order.
 * stats step : slab partials (sums around a pivot row) are combined by the LAST-ARRIVING block of the same
 * launch (ticket counter, fixed summation order: deterministic), no separate finalize launch.
 * The host may all-gather (mean | var | count) across ranks between the two steps for SyncBatchNorm (one
 * coalesced message per layer for all groups) and merge with xas_bn_sync_merge.
 * act: 0 none, 1 relu, 2 leaky-relu(0.01).
 * ---------------------------------------------------------------------------------- */
size_t xas_bn_workspace_floats(long M, int C, int groups);
/* mean[g*out_stride + c], var_biased[g*out_stride + c] (out_stride >= C, multiple of 4 floats; C for dense [G][C]).
 * count_out != NULL: count_out[g*out_stride] = M / groups as float (the count slot of a packed SyncBatchNorm
 * message [mean | var | count]: pass mean = msg, var = msg + C, count_out = msg + 2C, out_stride = 2C + 4).
 * running_mean/var != NULL (rank-local norms): the running-statistic updates (momentum, unbiased variance
 * from `count`) are applied in the same launch; pass NULL for SyncBatchNorm (xas_bn_sync_merge updates them). */
int xas_bn_stats(const float* x, long M, int C, int groups, float* mean, float* var_biased, long out_stride,
                 float* count_out, float* workspace, float* running_mean, float* running_var, float momentum,
                 long count, void* stream);
/* Statistics from per-tile partial sums: partial is a [rows][C][2] matrix, (sum(v - pivot), sum((v - pivot)^2)) of
 * rows_per_group / (rows / groups) consecutive activation rows each, the first rows / groups partial rows belonging to
 * group 0 and so on; pivot [C] or NULL (= 0).  Outputs and running-statistic update as xas_bn_stats.
 * workspace: xas_bn_workspace_floats(rows, 2 * C, groups). */
int xas_bn_stats_from_partials(const float* partial, long rows, int C, int groups, long rows_per_group,
                               const float* pivot, float* mean, float* var_biased, long out_stride, float* count_out,
                               float* workspace, float* running_mean, float* running_var, float momentum, void* stream);
/* Batch-norm backward sums from per-tile partial sums: partial [rows][2][C] (sum dz | sum dz * xhat of the activation rows
 * behind each partial row; the first rows / groups partial rows belong to group 0, ...) -> sums [groups][2][C]; the local
 * parameter gradients are added into dbeta_acc / dgamma_acc (both or neither).  workspace:
 * xas_bn_workspace_floats(rows, 2 * C, groups). */
int xas_bn_bwd_sums_from_partials(const float* partial, long rows, int C, int groups, float* sums, float* workspace,
                                  float* dbeta_acc, float* dgamma_acc, void* stream);
/* SyncBatchNorm merge (torch/nn/modules/_functions.py SyncBatchNorm.forward: batch_norm_gather_stats_with_counts):
 * gathered [world][groups][msg_stride] with mean at +0, biased var at +C, count at +2C of each message;
 * count-weighted merge in double -> mean, var_biased [groups][C]; running statistics (may be NULL) updated once per
 * group in group order with the global count. */
int xas_bn_sync_merge(const float* gathered, int world, int groups, int C, long msg_stride, float* mean,
                      float* var_biased, float* running_mean, float* running_var, float momentum, void* stream);
/* out[c] = sum_m x[m][c]  (bias gradients: deconv_head.py:34, physique_network.py:17, discriminator.py:11);
 * workspace: xas_bn_workspace_floats(M, C, 1) */
int xas_col_sum(const float* x, long M, int C, float* out, float* workspace, void* stream);
/* acc[c] += column sums (bias gradients accumulated in place: nn.Conv2d / nn.Linear bias, physique_network.py:16,
 * discriminator.py:23).  workspace as xas_col_sum. */
int xas_col_sum_acc(const float* x, long M, int C, float* acc, float* workspace, void* stream);
/* y = act(gamma*(x-mean_g)*rsqrt(var_g+eps)+beta [+ residual]); mean / var_biased: [groups][C].
 * mask_out (may be NULL): [M*C/4] bytes, bit e of byte i = (pre-activation value of element 4i+e > 0): all the backward
 * needs of y for a layer with a residual, at 1/16 of y's bytes. */
int xas_bn_apply(const float* x, const float* mean, const float* var_biased, const float* gamma,
                 const float* beta, const float* residual, float eps, int act, long M, int C, int groups,
                 float* y, uint8_t* mask_out, void* stream);
/* Same, and max |y| over the whole result is merged into the amax slot amax_out (XAS_AMAX_SLOT_FLOATS floats, zeroed by the
 * caller; atomic maximum on the bit pattern): the scale of y as the input of the next convolution's f16x3 launches (xas_conv_shape.grad_amax; x_amax of
 * its weight gradient).  amax_out == NULL: xas_bn_apply. */
int xas_bn_apply_amax(const float* x, const float* mean, const float* var_biased, const float* gamma,
                      const float* beta, const float* residual, float eps, int act, long M, int C, int groups,
                      float* y, uint8_t* mask_out, float* amax_out, void* stream);
/* max |x| over n floats merged into the amax slot amax_out (as above): for convolution inputs that no kernel of this library wrote
 * (images, rendered masks).  One streaming read. */
int xas_abs_max(const float* x, long n, float* amax_out, void* stream);
/* running = (1-momentum)*running + momentum*stat_g for g = 0..groups-1; var uses the unbiased estimate n/(n-1). */
int xas_bn_update_running(const float* mean, const float* var_biased, float* running_mean,
                          float* running_var, float momentum, long count, int C, int groups, void* stream);
/* backward, step 1: dz = dy * act'(y); sums[g][0][c] = sum_dz, sums[g][1][c] = sum_dz_xhat  ([groups][2][C]).
 * x may be NULL when the layer has an activation and no residual: xhat is then recovered from the saved output,
 * xhat = (act^-1(y) - beta) / gamma (needed only where dz != 0), one activation tensor less to read per pass.
 * y may be NULL instead (x, gamma, beta given, no residual): the activation mask is then re-derived from x with the
 * forward's exact arithmetic and the backward never reads y.
 * dbeta_acc / dgamma_acc (both or neither, may be NULL): the parameter gradients (sums over all groups) are ALSO
 * added into these [C] buffers (the .grad arena of the optimizer), which saves the autograd accumulation kernels. */
int xas_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                      const float* var_biased, const float* gamma, const float* beta, float eps, int act,
                      long M, int C, int groups, float* sums, float* workspace,
                      float* dbeta_acc, float* dgamma_acc, const uint8_t* mask, void* stream);
/* mask (may be NULL): the sign bytes xas_bn_apply wrote; when given, y is not read (x required). */
/* step 2: dx = gamma*invstd_g*(dz - sum_dz_g/cnt - xhat*sum_dz_xhat_g/cnt); dres = dz if != NULL; cnt = rows per
 * group (x world size for SyncBatchNorm).  x may be NULL only for the leaky-ReLU layers (invertible activation). */
int xas_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                     const float* var_biased, const float* gamma, const float* beta, const float* sums,
                     float eps, int act, long M, int C, int groups, double count,
                     float* dx, float* dresidual, const uint8_t* mask, void* stream);
/* Same, and max |dx| over the whole result is merged into the amax slot amax_out (XAS_AMAX_SLOT_FLOATS device floats, zeroed
 * by the caller before the first launch that merges into it; atomic maximum on the bit pattern): the scale of dx as the gradient operand of the f16x3
 * data / weight gradients (xas_conv_shape.grad_amax).  amax_out == NULL: xas_bn_bwd_apply. */
int xas_bn_bwd_apply_amax(const float* x, const float* y, const float* dy, const float* mean,
                          const float* var_biased, const float* gamma, const float* beta, const float* sums,
                          float eps, int act, long M, int C, int groups, double count, float* dx,
                          float* dresidual, const uint8_t* mask, float* amax_out, void* stream);

/* 3x3 stride-2 pad-1 max pool (resnet.py:20), NHWC. idx: int8 argmax tap 0..8 for backward */
int xas_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int8_t* idx, void* stream);
int xas_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, int N, int H, int W, int C, float* dx, void* stream);

/* bilinear x2 upsample, align_corners=False (physique_network.py:31), NHWC */
int xas_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream);
int xas_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream);

/* elementwise: y = sigmoid(x) ; dx = dy*y*(1-y) */
int xas_sigmoid_fwd(const float* x, long n, float* y, void* stream);
int xas_sigmoid_bwd(const float* y, const float* dy, long n, float* dx, void* stream);
/* NCHW -> NHWC and back (input images arrive NCHW: dataloader.py:166) */
int xas_nchw_to_nhwc(const float* x, int N, int C, int H, int W, float* y, void* stream);
int xas_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y, void* stream);

/* ------------------------------------------------------------------------------------
 * GPU input pipeline: the per-sample work of the reference's CPU loader
 * (human_utils/dataloader/dataloader.py:17-91,150-191; OpenCV + scikit-fmm) over a batch of decoded 8-bit images in HBM.
 * ---------------------------------------------------------------------------------- */
/* cv2.warpAffine(src_b, trans_b, (P, P), flags=INTER_LINEAR) with BORDER_CONSTANT 0 for every image of the batch
 * (common/imglib/affine.py:112, dataloader.py:58).  src: packed 8-bit images [H_b][W_b][C] at byte offsets src_off[b];
 * src_hw [B][2] = (H_b, W_b); minv [B][6] doubles = the INVERTED 2x3 transform (patch pixel -> source pixel), inverted on
 * the host as cv::warpAffine does; dst [B][P][P][C].  OpenCV's fixed-point arithmetic, bit-exact. */
int xas_warp_affine_u8(const uint8_t* src, const long* src_off, const int* src_hw, const double* minv, int B, int C,
                       int P, uint8_t* dst, void* stream);
/* cv2.GaussianBlur(mask, (5,5), 0) then cv2.threshold(127, 255, THRESH_BINARY) (MPI-INF-3DHP masks, dataloader.py:62-65);
 * mask, out: [B][P][P] 8-bit, out != mask */
int xas_mask_blur_threshold(const uint8_t* mask, int B, int P, uint8_t* out, void* stream);
/* convert_cvimg_to_tensor + colour scale / clip + (x - mean) / std + mask / 255 + rm_bg (dataloader.py:56,67-71,185-188):
 * img_bgr [B][P][P][3], mask [B][P][P] 8-bit -> out_img [B][3][P][P] RGB float, out_mask [B][1][P][P] float.
 * color_scale: device [B][3] (RGB order) or NULL; mean3 / std3: HOST pointers to 3 floats (RGB order). */
int xas_patch_finish(const uint8_t* img_bgr, const uint8_t* mask, const float* color_scale, const float* mean3,
                     const float* std3, int rm_bg, int B, int P, float* out_img, float* out_mask, void* stream);
/* compute_geodesic_dis (common/utility/geodesic.py:14-54): mask [B][P][P] float (non-zero = foreground);
 * centers: device [B][2] ints (x, y) or NULL (centroid of the mask, geodesic.py:4-12); params5: HOST pointer to
 * geodesic_param_list (5 floats, [4] must be 0); out [B][1][P][P]; center_out: device [B][2] ints or NULL.
 * Grid Eikonal solves by iterating the first-order upwind update to its fixed point (scikit-fmm: fast marching). */
size_t xas_geodesic_workspace_bytes(int B, int P);
int xas_geodesic_weight(const float* mask, const int* centers, const float* params5, int B, int P, float* out,
                        int* center_out, void* workspace, void* stream);
/* The general form: SEVERAL source pixels per image (geodesic_pt_list: the listed joints, dataloader.py:189-191) and the order
 * of the upwind scheme.  centers: device [B][num_centers][2] ints (x, y), 1 <= num_centers <= 64, or NULL with num_centers = 1
 * (centroid); every centre is a zero of the inside solve; if ANY centre of an image lies on the background that image's map is
 * all ones (geodesic.py:22-27).  order: 2 = the second-order scheme scikit-fmm's `distance` runs by default (one-sided
 * second-order differences where the second upwind neighbour is not larger, first order elsewhere; restated from the
 * library's documented algorithm, parity unpinned), 1 = first order (what xas_geodesic_weight computes).  center_out: device
 * [B][2] ints (first centre) or NULL.  Workspace as above. */
int xas_geodesic_weight_multi(const float* mask, const int* centers, int num_centers, int order, const float* params5, int B, int P,
                              float* out, int* center_out, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Mask losses (modules/base_losses/loss_func.py:4-16), fused clip * weight * MSE.
 * mode bit0: use_clip (mask > 0.1), bit1: has weight.
 * out: 3 floats.
 *   weight given : out[0] = mean(e * clip * weight), out[1] = 1
 *   no weight    : out[0] = mean((m-gt)^2), out[1] = mean(clip)   (the reference returns the
 *                  tensor mse*clip; its later .mean() equals out[0]*out[1])
 *   out[2] = out[0]*out[1] = the scalar the trainer reduces to (train.py:182)
 * partial: workspace of xas_loss_nblk(n) * 2 floats.
 * bwd: dm = grad_scalar * d(out)/dm for the weighted form; for the unweighted+clip form
 *      dm = grad_scalar * out[1] * 2(m-gt)/n.
 * ---------------------------------------------------------------------------------- */
int xas_loss_nblk(long n);
int xas_mask_loss_fwd(const float* m, const float* gt, const float* weight, long n, int mode,
                      float* partial, float* out, void* stream);
int xas_mask_loss_bwd(const float* m, const float* gt, const float* weight, long n, int mode,
                      const float* out, const float* grad_scalar, float* dm, void* stream);

/* ------------------------------------------------------------------------------------
 * Joint-level losses as the model combines them (loss_func.py:18-76, model.py:98-164): a batch mean per
 * hypothesis, then the MIN over hypotheses; gradient flows to the winning hypothesis only.
 * pred [B][Hy][K][3].  kind 0: supervision vs gt [B][K][3] ; kind 1: w0*bone_sym + w1*kp_sym (3-D, mm*1e-3)
 * [+ w2 * kp_sym on the (x,y) of the patch joints passed as `gt` = kps [B][Hy][K][3], or NULL] ;
 * kind 2: w0 * kp_sym on (x,y).  out: 2+Hy floats = {min, argmin, v_0..v_{Hy-1}}.  grad_aux (kind 1 with the
 * 2-D term, else NULL): gradient w.r.t. the patch joints.
 * xas_lsgan_*: logits [B][Hy] -> out[0] = mean_b min_h (x - target)^2, idx[b] = argmin.
 * ---------------------------------------------------------------------------------- */
int xas_pose_loss_fwd(const float* pred, const float* gt, int B, int Hy, int K, int kind, float w0, float w1,
                      float w2, float* out, void* stream);
int xas_pose_loss_bwd(const float* pred, const float* gt, int B, int Hy, int K, int kind, float w0, float w1,
                      float w2, const float* out, const float* grad_scalar, float* grad_pred, float* grad_aux,
                      void* stream);
int xas_lsgan_fwd(const float* logits, int B, int Hy, float target, float* out, int* idx, void* stream);
int xas_lsgan_bwd(const float* logits, int B, int Hy, float target, const int* idx, const float* grad_scalar,
                  float* grad_logits, void* stream);

/* ------------------------------------------------------------------------------------
 * GCN discriminator building blocks (discriminator.py:180-238, gcn.py:79-110 with
 * torch_geometric SAGEConv(mean) / graph LayerNorm semantics).  Node features [B*N][C].
 * ---------------------------------------------------------------------------------- */
/* y[b,i,:] = sum_j adj[i][j] * x[b,j,:]  (adj: N x N row-normalised, device) */
int xas_graph_aggregate(const float* x, const float* adj, int B, int N, int C, float* y, void* stream);
/* graph LayerNorm (normalise over the WHOLE [rows, C] tensor of one discriminator call) + per-channel
 * affine + relu (+ residual).  `groups` independent calls are batched: x is [groups*rows, C] and the
 * statistics are per group.  stats: [groups][2] = (mean, std).
 * workspace: xas_gln_workspace_floats(rows*C, groups, C) floats. */
size_t xas_gln_workspace_floats(long n, int groups, int C);
int xas_gln_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                long rows, int C, int groups, float eps, float* y, float* stats, float* workspace,
                void* stream);
/* backward of y = relu(ln(x)) (+ residual: pass-through handled by the caller); the relu mask is
 * recomputed from x, gamma, beta; dgamma / dbeta are summed over the groups. */
int xas_gln_bwd(const float* x, const float* beta, const float* dy, const float* gamma,
                const float* stats, long rows, int C, int groups, float eps, float* dx, float* dgamma,
                float* dbeta, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * SMPL linear blend skinning (modules/smplpytorch/pytorch/smpl_layer.py:63-156).
 * ---------------------------------------------------------------------------------- */
int xas_smpl_lbs_fwd(const float* pose /*[B][72]*/, const float* betas /*[B][10]*/,
                     const float* v_template /*[V][3]*/, const float* shapedirs /*[V][3][10]*/,
                     const float* posedirs /*[V][3][207]*/, const float* j_regressor /*[24][V]*/,
                     const float* weights /*[V][24]*/, const int* parents /*[24]*/, int B, int V,
                     int center_idx, float* verts /*[B][V][3]*/, float* joints /*[B][24][3]*/,
                     float* workspace /* B*(24*3 + 24*16 + 207) floats */, void* stream);

/* Backward of xas_smpl_lbs_fwd (autograd of smpl_layer.py:63-156): d_verts [B][V][3], d_joints [B][24][3] (or NULL)
 * -> d_pose [B][72], d_betas [B][10].  fwd_workspace: the workspace the forward call filled for the same inputs.
 * workspace: xas_smpl_lbs_bwd_workspace_floats(B, V).  Deterministic (no atomics). */
size_t xas_smpl_lbs_bwd_workspace_floats(int B, int V);
int xas_smpl_lbs_bwd(const float* pose, const float* betas, const float* v_template, const float* shapedirs,
                     const float* posedirs, const float* j_regressor, const float* weights, const int* parents, int B,
                     int V, int center_idx, const float* fwd_workspace, const float* d_verts, const float* d_joints,
                     float* d_pose, float* d_betas, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-tensor Adam (train.py:257-264: betas (0.5, 0.999), eps 1e-8, no decay).
 * One launch updates a flat parameter arena: p, g, m, v are arenas of n floats.
 * ---------------------------------------------------------------------------------- */
int xas_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream);

/* ------------------------------------------------------------------------------------
 * Evaluation path: hypothesis selection, triangulation, pose metrics.
 * One evaluation batch stays on the device; nothing here synchronises with the host.
 * ---------------------------------------------------------------------------------- */

/* Left/right switch + hypothesis selection + 2-D error for ONE camera, one launch.
 * Replaces eval_utils.py:7-30 (switch_points), eval.py:128-148 (per-hypothesis loop, best / confident
 * selection by argmin + gather) and eval_utils.py:32-43 (per_act_mse).
 * kps [B][Hy][K][C] (C = 2 or 3), joints [B][K][C] ground truth in PIXEL units unless
 * XAS_EVAL_GT_NORMALISED; perm [K] = index of the mirrored joint (identity where none).
 * Outputs (each may be NULL): sel3d [B][K][C] (selection by the C-dim squared error),
 * sel2d [B][K][2] (selection by the 2-D squared error), err2d [B], swapped [B][K] (0/1, flags of the LAST
 * hypothesis - what eval.py:132 keeps).  K <= 64. */
#define XAS_EVAL_CONFIDENT 1      /* take hypothesis 0 (eval.py:144-146) instead of the best one */
#define XAS_EVAL_SWITCH_ALL 2     /* one switch decision per sample (switch_points(switch_all=True)) */
#define XAS_EVAL_GT_NORMALISED 4  /* joints already in the detector's output range */
int xas_eval_select(const float* kps, const float* joints, const int* perm, int B, int Hy, int K, int C,
                    float image_size, int flags, float* sel3d, float* sel2d, float* err2d,
                    unsigned char* swapped, void* stream);

/* P [B][3][4] = k_mat [B][3][3] * [rot_world [B][3][3] | trans_world [B][3]].  modules/util.py:188. */
int xas_projection_matrix(const float* k_mat, const float* rot_world, const float* trans_world, int B, float* P,
                          void* stream);

/* Weighted DLT triangulation, replaces modules/util.py:198-230 (batch_triangulate: torch.linalg.svd of
 * [B][K][2V][4]).  points [B][V][K][3] = (u, v, weight) image coordinates, P [B][V][3][4];
 * out [B][K][4] = (X/w, Y/w, Z/w, mean weight).  V >= 2. */
int xas_triangulate_dlt(const float* points, const float* P, int B, int V, int K, float* out, void* stream);

/* Pose metrics, replaces metrics.py:5-244 (numpy, per-sample SVD on the host).
 * pred, gt [N][K][3]; both are divided by in_div on load (eval.py:52-53 passes mm / 1000 for PCK / AUC);
 * mask [N][K] (0/1) or NULL = all visible.  Outputs (each may be NULL):
 *   err [3][N][K]       per-joint error * mask under alignment none / scale / procrustes
 *   aligned [2][N][K][3] the scale- and procrustes-aligned predictions
 *   pck [N][K]          100 where err[pck_align] < pck_threshold, * mask
 *   auc_hits [N][31]    visible joints with err[pck_align] < linspace(0, 0.15, 31)[t]
 * 3 <= K <= 64. */
int xas_pose_metrics(const float* pred, const float* gt, const unsigned char* mask, int N, int K, float in_div,
                     int pck_align, float pck_threshold, float* err, float* aligned, float* pck, int* auc_hits,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
