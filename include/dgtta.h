/*
 * dgtta.h — C ABI of libdgtta_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * DG-TTA test-time-adaptation hot path.
 *
 * The reference (multimodallearning/DG-TTA) is pure Python on stock PyTorch ops and has no FFI of
 * its own; this ABI is the drop-in boundary that sits UNDER the reference's Python operator
 * signatures (dg_tta_amd/ mirrors those).  Each entry point cites the reference code it replaces
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless named h_*.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns;
 *    nothing here allocates, frees or synchronises, so calls may be captured into a hipGraph.
 *  - the caller owns all memory, including workspaces (sizes from the *_ws_bytes queries).
 *  - return value: 0 = OK, <0 = DGTTA_ERR_*; dgtta_last_error() gives a thread-local message.
 *  - volumes are D x H x W (W fastest).  Layouts:
 *        NCDHW  [B][C][D][H][W]          (PyTorch contiguous; the reference's layout)
 *        NDHWC  [B][D][H][W][ldc]        (channels-last; `ldc` >= C elements per voxel, so a
 *                                          tensor may be a channel slice of a wider buffer)
 *  - dtype: 0 = fp32, 1 = bf16, 2 = fp16 (storage of activations and packed weights; accumulation, statistics,
 *    weight gradients and the optimizer state are always fp32).  fp16 gradients need the caller's static loss scale
 *    (dgtta_adamw_step divides it out).
 */
#ifndef DGTTA_H
#define DGTTA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGTTA_OK 0
#define DGTTA_ERR_BADARG (-1)
#define DGTTA_ERR_UNSUPPORTED (-2)
#define DGTTA_ERR_WORKSPACE (-3)
#define DGTTA_ERR_LAUNCH (-4)

#define DGTTA_F32 0
#define DGTTA_BF16 1
#define DGTTA_F16 2

#define DGTTA_PAD_ZEROS 0
#define DGTTA_PAD_BORDER 1
#define DGTTA_INTERP_LINEAR 0
#define DGTTA_INTERP_NEAREST 1

int dgtta_version(void);
const char *dgtta_last_error(void);
/* The DGTTA_* diagnostic switches are read from the environment once, at first use; this call takes a
 * fresh snapshot (tests that flip a switch inside one process call it after changing the environment). */
int dgtta_reload_env(void);

/* ---------------------------------------------------------------------------------------------
 * MIND3D descriptor.  Replaces MIND3D.forward + smooth + filter1D (dg_tta/mind.py:142-164, :27-43,
 * :5-24) and mind_hook (:167-168).  img [B,1,D,H,W] fp32; noise [B,12,D,H,W] fp32 = the
 * torch.randn_like draw of mind.py:150 (caller supplies it so RNG semantics stay the caller's).
 * out: 12 channels, either NCDHW fp32 (out_ndhwc=0) or NDHWC with row length out_ldc (>=12;
 * channels 12..out_ldc-1 are written as zero) in out_dtype.  delta (mind.py:98,137: neighbour distance,
 * 1 or 2) and the Gaussian taps of `sigma` (h_taps: HOST array of 3 / 5 / 7 floats evaluated as mind.py:30-37
 * does, i.e. sigma <= 2; the reference itself only ever uses delta = 1, sigma = 1).
 * ws: dgtta_mind3d_ws_bytes(B,D,H,W) bytes.
 * ------------------------------------------------------------------------------------------- */
size_t dgtta_mind3d_ws_bytes(int B, int D, int H, int W);
int dgtta_mind3d_fwd(const float *img, const float *noise, float randn_weighting, int delta, const float *h_taps,
                     int ntaps, void *out, int out_ndhwc, int out_ldc, int out_dtype, void *ws, size_t ws_bytes, int B,
                     int D, int H, int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GIN random-convolution chain.  Replaces GINGroupConv.forward / GradlessGCReplayNonlinBlock.forward
 * (dg_tta/gin.py:168-230, :59-122) for gin_aug's fixed config (1->2->2->2->1 channels, 4 layers).
 * x,out [B,1,D,H,W] fp32.  alpha [B].  ksz[4] HOST ints in {1,3}.  ker[l] = device pointer to
 * [B*cout_l, cin_l, k,k,k] fp32 exactly as drawn at gin.py:94-97; shift[l] = [B*cout_l] (gin.py:98-103).
 * ws: dgtta_gin_ws_bytes(B,D,H,W).
 * ------------------------------------------------------------------------------------------- */
size_t dgtta_gin_ws_bytes(int B, int D, int H, int W);
int dgtta_gin_chain_fwd(const float *x, const float *alpha, const int *h_ksz, const float *const *h_ker,
                        const float *const *h_shift, float *out, void *ws, size_t ws_bytes, int B, int D, int H,
                        int W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Affine resampling = F.affine_grid + F.grid_sample(align_corners=False) without materialising
 * the grid.  Replaces the call sites dg_tta/tta/tta.py:523-551 (image, border), :572-575 (logits,
 * zeros) and dg_tta/tta/torch_utils.py:55-73 (get_batch; zeros; linear / nearest).
 * theta [B,3,4] fp32 device (x,y,z order of F.affine_grid).  If tta_grid_algebra != 0 the sampling
 * grid is formed as ((affine_grid(theta) - identity) + identity) in fp32, as tta.py:523-548 does.
 * src [B,C,Ds,Hs,Ws] / dst [B,C,Dd,Hd,Wd], layout NCDHW (ndhwc=0) or NDHWC with row lengths
 * src_ldc/dst_ldc.  fp32 only.  If sub_const_dev != NULL the scalar it points to is subtracted
 * before and added back after sampling (get_batch's "(img - img_min) ... + img_min", torch_utils.py:55-73).
 * bwd: grad_src = d(dst)/d(src)^T grad_dst (linear only; theta gets no gradient, as in the
 * reference where R is a constant).  grad_src is overwritten; it needs no initialisation.
 * ------------------------------------------------------------------------------------------- */
int dgtta_affine_warp3d_fwd(const float *src, const float *theta, float *dst, int B, int C, int Ds, int Hs,
                            int Ws, int Dd, int Hd, int Wd, int ndhwc, int src_ldc, int dst_ldc, int pad_mode,
                            int interp_mode, int tta_grid_algebra, const float *sub_const_dev, void *stream);
int dgtta_affine_warp3d_bwd(const float *grad_dst, const float *theta, float *grad_src, int B, int C, int Ds,
                            int Hs, int Ws, int Dd, int Hd, int Wd, int ndhwc, int src_ldc, int dst_ldc,
                            int pad_mode, int tta_grid_algebra, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Consistency loss.  Replaces dg_tta/tta/tta.py:263-271 + soft_dice_loss (torch_utils.py:90-104):
 * mask=(sum_c a>0)(sum_c b>0); sm=softmax_c * mask; dice_c = mean(2ab)/mean((a+b)^2/2);
 * loss = 1 - mean_{b, c>=start_class} dice.  la, lb: logits NDHWC [B][V][ldc], C classes, fp32.
 * fwd writes dice[B*C], loss[1] and keeps what bwd needs in ws.  bwd writes grad_la/grad_lb
 * (same layout) = grad_scale * (grad_scale_dev ? *grad_scale_dev : 1) * dloss/dlogits (mask treated as
 * a constant, as autograd does; the device scalar lets autograd's upstream gradient stay on the GPU).
 * guard_items: soft_dice_loss's "denominator.sum() == 0 -> dice = 1" guard is evaluated per group of
 * guard_items consecutive batch items (B = one reference call; 1 = every item is its own call, used
 * when several accumulation steps of batch size 1 run as one batch).  Must divide B.
 * ------------------------------------------------------------------------------------------- */
size_t dgtta_softdice_ws_bytes(int B, int C, int64_t V);
int dgtta_softdice_fwd(const float *la, const float *lb, float *dice, float *loss, void *ws, size_t ws_bytes,
                       int B, int C, int64_t V, int ldc, int start_class, int guard_items, void *stream);
int dgtta_softdice_bwd(const float *la, const float *lb, float *grad_la, float *grad_lb, const void *ws,
                       float grad_scale, const float *grad_scale_dev, int B, int C, int64_t V, int ldc,
                       int start_class, void *stream);
/* dgtta_softdice_bwd with the gradient rows in `grad_dtype` (DGTTA_F32: the call above; DGTTA_BF16 / DGTTA_F16: the 16-class
 * form only - C = ldc = 16 - each value rounded once on the way out; what dgtta_seghead_warp_bwd_g16 consumes). */
int dgtta_softdice_bwd_t(const float *la, const float *lb, void *grad_la, void *grad_lb, const void *ws, float grad_scale,
                         const float *grad_scale_dev, int B, int C, int64_t V, int ldc, int start_class, int grad_dtype,
                         void *stream);

/* Stand-alone soft_dice_loss(smp_a, smp_b) -> dice[B][C] of the reference (dg_tta/tta/torch_utils.py:90-104) on
 * probability maps: nom = mean_v(2ab), den = mean_v((a+b)^2)/2, dice = nom/den, all ones when den.sum() == 0.
 * a, b: fp32 [B][C][V] addressed with element strides (stride_b, stride_c, stride_v): NCDHW has stride_v = 1,
 * channels-last stride_c = 1.  bwd: grad_a / grad_b (same strides) = sum_c grad_dice[b][c] * d dice / d a.
 * ws: dgtta_softdice_ws_bytes(B, C, V), kept between fwd and bwd. */
int dgtta_softdice_probs_fwd(const float *a, const float *b, float *dice, void *ws, size_t ws_bytes, int B, int C,
                             int64_t V, int64_t stride_b, int64_t stride_c, int64_t stride_v, void *stream);
int dgtta_softdice_probs_bwd(const float *a, const float *b, const float *grad_dice, float *grad_a, float *grad_b,
                             const void *ws, int B, int C, int64_t V, int64_t stride_b, int64_t stride_c,
                             int64_t stride_v, void *stream);

/* Supervised loss of the PRE-TRAINING side: soft Dice + cross-entropy against an integer label map -
 * nnU-Net's DC_and_CE_loss [3P nnunetv2==2.2.1], the loss of the nnUNetTrainer the reference's trainers inherit
 * (dg_tta/pretraining/nnUNetTrainer_GIN.py:38-59, nnUNetTrainer_MIND.py:37-57, nnUNetTrainer_GIN_MIND.py:38-59):
 *   p = softmax_c(logits);  voxels whose label is outside [0, C) are ignored
 *   loss3[0] = loss3[1] + loss3[2];  loss3[1] = -mean_v log p[v][y_v];  loss3[2] = -mean_{b, c >= (do_bg ? 0 : 1)} dice[b][c]
 *   dice[b][c] = (2 sum_v p y + smooth) / (sum_v p + sum_v y + smooth)          (per sample: batch_dice = False)
 * logits / grad_logits: voxel-major fp32 [B][V][ld], labels int64 [B][V].  bwd: grad_logits = grad_scale * (*grad_scale_dev
 * if given) * d loss3[0] / d logits.  ws: dgtta_dice_ce_ws_bytes(B, C, V), kept between fwd and bwd.  1 <= B <= 8, C <= 128. */
size_t dgtta_dice_ce_ws_bytes(int B, int C, int64_t V);
int dgtta_dice_ce_fwd(const float *logits, int ldc, const int64_t *labels, float *loss3, float *dice, void *ws,
                      size_t ws_bytes, int B, int C, int64_t V, float smooth, int do_bg, void *stream);
int dgtta_dice_ce_bwd(const float *logits, int ldc, const int64_t *labels, const void *ws, float grad_scale,
                      const float *grad_scale_dev, float *grad_logits, int ldg, int B, int C, int64_t V, void *stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW (decoupled weight decay, bias-corrected, no amsgrad) over a list of tensors.  Replaces
 * torch.optim.AdamW(model.parameters(), lr).step() at dg_tta/tta/tta.py:185,278 (betas 0.9/0.999,
 * eps 1e-8, weight_decay 0.01 = PyTorch defaults).  h_* are HOST arrays of ntensors device
 * pointers / element counts; tensors whose h_g[i] is NULL are skipped (grad None in PyTorch).
 * step is the 1-based step count of those tensors.  grad_scale: the static loss scale the gradients carry (fp16
 * storage path: the caller multiplied the loss gradient by it); every gradient is divided by it on load. 1 = none.
 * ------------------------------------------------------------------------------------------- */
int dgtta_adamw_step(float *const *h_p, const float *const *h_g, float *const *h_m, float *const *h_v,
                     const int64_t *h_n, int ntensors, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, const int *skip_if_nonzero, void *stream);
/* Overflow guard of the fp16 storage path (the role torch.cuda.amp.GradScaler's inf check plays for autocast users; the
 * reference itself runs fp32, dg_tta/tta/tta.py:275-279): *flag (device int, zeroed by the caller) becomes 1 when any
 * gradient element is inf / NaN.  dgtta_adamw_step(skip_if_nonzero = flag) then leaves parameters and state untouched;
 * skip_if_nonzero may be NULL. */
int dgtta_grads_nonfinite(const float *const *h_g, const int64_t *h_n, int ntensors, int *flag, void *stream);

/* ---------------------------------------------------------------------------------------------
 * nnUNet PlainConvUNet building blocks (third-party dynamic-network-architectures==0.2, built at
 * dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53 from plans.json:279-401): Conv3d 3x3x3 (pad 1,
 * stride 1|2, bias) -> InstanceNorm3d(eps, affine) -> LeakyReLU(0.01); ConvTranspose3d k2 s2;
 * 1x1x1 segmentation head.  Activations are NDHWC with explicit row lengths (ld*), so that
 * torch.cat((up, skip), 1) is realised by writing into channel slices of one buffer.
 * Weights: w_t = torch layout [Cout][Cin][3][3][3] fp32 (read-only); the kernels use packed copies
 * produced by dgtta_conv3d_pack_weights.
 * ------------------------------------------------------------------------------------------- */

/* packs torch [Cout][Cin][27] fp32 into one blob `wpack` of 2*27*CinP*CoutP elements (dtype fp32|bf16, zero padded):
 *   wf [27][CinP][CoutP] = w[co][ci][tap]        (first half)
 *   wb [27][CoutP][CinP] = w[co][ci][26-tap]     (second half: taps mirrored, channel roles swapped)
 * forward, data-gradient and MFMA kernels pick the orientation they need from the blob. */
size_t dgtta_conv3d_packed_bytes(int CinP, int CoutP, int dtype);
int dgtta_conv3d_pack_weights(const float *w_t, void *wpack, int Cin, int Cout, int CinP, int CoutP, int dtype,
                              void *stream);

/* y[b][vo][co] = bias[co] + sum_{tap,ci} x[b][s*vo+tap-1][ci] * w[co][ci][tap]      (zero padding)
 * Optionally accumulates per-(b,co) partial sums of y and y^2 for the following InstanceNorm
 * (stats != NULL: dgtta_conv3d_stats_bytes).  impl: 0 = auto, 1 = reference-grade VALU kernel,
 * 2 = MFMA implicit GEMM. */
size_t dgtta_conv3d_stats_bytes(int B, int Cout, int Do, int Ho, int Wo);
int dgtta_conv3d_k3_fwd(const void *x, int ldx, const void *wpack, const float *bias, void *y, int ldy, void *stats,
                        int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int stride,
                        int dtype, int impl, void *stream);
/* dx[b][vi][ci] = sum_{tap,co} dy[b][(vi+1-tap)/s][co] * w[co][ci][tap]   (terms with non-integer
 * or out-of-range index dropped).  accumulate != 0: dx += (skip-connection gradient sum). */
int dgtta_conv3d_k3_dgrad(const void *dy, int lddy, const void *wpack, void *dx, int lddx, int B, int Cin, int Cout,
                          int CinP, int CoutP, int Di, int Hi, int Wi, int stride, int accumulate, int dtype,
                          int impl, void *stream);
/* dw_t[co][ci][tap] (+)= sum_{b,vo} x[b][s*vo+tap-1][ci] * dy[b][vo][co];  db[co] (+)= sum dy.
 * fp32 outputs in torch layout; accumulate != 0 adds to existing gradients (gradient accumulation
 * over tta.py:221's 16 steps and over both branches). */
size_t dgtta_conv3d_wgrad_ws_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo);
/* Optional larger workspace for dtype = DGTTA_F32 (stride 1, or 2 with even extents): the plain workspace followed by room for
 * three bf16 planes of x (input extent) and three of dy.  Given that much, dgtta_conv3d_k3_wgrad evaluates the fp32 weight gradient as six launches of the 16-bit
 * matrix-core kernels on the exact three-term bf16 splits of its operands (x = x0 + x1 + x2; the six products with i + j <= 2,
 * each exact in the fp32 accumulator) instead of the fp32 MFMA kernel: same result to fp32 rounding, ~1.7x the rate. */
size_t dgtta_conv3d_wgrad_split_ws_bytes(int B, int Cin, int Cout, int Do, int Ho, int Wo, int stride);
int dgtta_conv3d_k3_wgrad(const void *x, int ldx, const void *dy, int lddy, float *dw_t, float *db, void *ws,
                          size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int stride,
                          int accumulate, int dtype, int impl, void *stream);
/* Round 6: x as 32-channel BLOCKS.  Block c of the input channels is a dense tensor [B][D][H][W][32] at element offset
 * c * x_block_stride from x (the level-0 concat buffer of the U-Net as planes [up | skip]: the kernels that read ONE half - the
 * stride-2 conv of the skip, the transposed conv's backward - then use whole 128-byte lines).  Stride 1, 64 input channels, 16-bit
 * storage, and only where the D-ring kernels take the launch: dgtta_conv3d_k3_blocked_supported returns 1 exactly when BOTH calls below
 * accept these dimensions; otherwise they return DGTTA_ERR_UNSUPPORTED and launch nothing.  Results equal the interleaved
 * layout's bit for bit (same kernels, same order). */
int dgtta_conv3d_k3_blocked_supported(int B, int Cin, int Cout, int D, int H, int W, int dtype);
int dgtta_conv3d_k3_fwd_blocked(const void *x, long long x_block_stride, const void *wpack, const float *bias, void *y, int ldy,
                                void *stats, int B, int Cin, int Cout, int CinP, int CoutP, int Di, int Hi, int Wi, int dtype,
                                void *stream);
int dgtta_conv3d_k3_wgrad_blocked(const void *x, long long x_block_stride, const void *dy, int lddy, float *dw_t, float *db,
                                  void *ws, size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int accumulate,
                                  int dtype, void *stream);

/* InstanceNorm3d(eps, affine, biased variance) + LeakyReLU(slope), per (b,c) over the volume.
 * fwd: stats = partial sums from the conv epilogue, or NULL (then computed here from y).
 *      writes mean_rstd[B][C][2] (saved for backward) and z = lrelu(gamma*(y-mean)*rstd + beta). */
size_t dgtta_instnorm_ws_bytes(int B, int C, int64_t V);
int dgtta_instnorm_lrelu_fwd(const void *y, int ldy, const void *stats, const float *gamma, const float *beta,
                             float *mean_rstd, void *z, int ldz, void *ws, size_t ws_bytes, int B, int C,
                             int64_t V, float eps, float slope, int dtype, void *stream);
/* bwd: given gz = dL/dz and saved y, mean_rstd: writes dy (may alias gz), and dgamma/dbeta (+)=. */
int dgtta_instnorm_lrelu_bwd(const void *gz, int ldgz, const void *y, int ldy, const float *gamma,
                             const float *beta, const float *mean_rstd, void *dy, int lddy, float *dgamma,
                             float *dbeta, void *ws, size_t ws_bytes, int B, int C, int64_t V, float slope,
                             int accumulate, int dtype, void *stream);

/* Data gradient of a stride-1 conv whose INPUT was z = LeakyReLU(InstanceNorm(y_prev)) (the blocks built at
 * dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:46-53; autograd of tta.py:275), fused with the reduction pass of that
 * previous layer's InstanceNorm backward: besides dx = gz it leaves, in `gstats` (size dgtta_conv3d_stats_bytes(B, Cin,
 * Di, Hi, Wi)), the per-tile sums of g' and g' * y_prev (g' = gz * lrelu'(pre-activation)).  *h_produced (HOST int) = 1 when
 * the statistics were produced (16-bit storage on the row-reuse kernel, whole tiles); 0: plain dgtta_conv3d_k3_dgrad ran.
 * dgtta_instnorm_lrelu_bwd_gstats = dgtta_instnorm_lrelu_bwd without its own pass over y and gz. */
int dgtta_conv3d_k3_dgrad_gstats(const void *dy, int lddy, const void *wpack, void *dx, int lddx, int B, int Cin, int Cout,
                                 int CinP, int CoutP, int Di, int Hi, int Wi, const void *y_prev, int ldy_prev,
                                 const float *mean_rstd_prev, const float *gamma_prev, const float *beta_prev, float slope,
                                 void *gstats, size_t gstats_bytes, int *h_produced, int dtype, int impl, void *stream);
int dgtta_instnorm_lrelu_bwd_gstats(const void *gz, int ldgz, const void *y, int ldy, const float *gamma, const float *beta,
                                    const float *mean_rstd, void *dy, int lddy, float *dgamma, float *dbeta,
                                    const void *gstats, void *ws, size_t ws_bytes, int B, int C, int64_t V, float slope,
                                    int accumulate, int dtype, void *stream);

/* ConvTranspose3d(Cin, Cout, kernel 2, stride 2, bias): w_t torch layout [Cin][Cout][2][2][2] fp32.
 * out[b][2v+o][co] = bias[co] + sum_ci x[b][v][ci] * w[ci][co][o]. */
size_t dgtta_convT3d_fwd_ws_bytes(int Cin, int Cout, int dtype);
int dgtta_convT3d_k2s2_fwd(const void *x, int ldx, const float *w_t, const float *bias, void *out, int ldo, void *ws,
                           size_t ws_bytes, int B, int Cin, int Cout, int Di, int Hi, int Wi, int dtype, int impl,
                           void *stream);
size_t dgtta_convT3d_bwd_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi);
/* (round 5) the same with room for the fp32 weight gradient as six launches of the 16-bit matrix-core kernel on the exact three-term
 * bf16 splits of x and dout (see dgtta_conv3d_wgrad_split_ws_bytes): offered this workspace, dgtta_convT3d_k2s2_bwd takes that path in
 * fp32 storage; with the plain one it runs the fp32 MFMA kernel as before. */
size_t dgtta_convT3d_bwd_split_ws_bytes(int B, int Cin, int Cout, int Di, int Hi, int Wi);
int dgtta_convT3d_k2s2_bwd(const void *x, int ldx, const void *dout, int lddo, const float *w_t, void *dx, int lddx,
                           float *dw_t, float *db, void *ws, size_t ws_bytes, int B, int Cin, int Cout, int Di,
                           int Hi, int Wi, int accumulate, int dtype, int impl, void *stream);

/* 1x1x1 head fused with map_label(input_format="logits") (torch_utils.py:214-221): only the rows
 * sel[0..nsel) of the [Ccls][Cin] weight are evaluated (sel == NULL: all Ccls rows).
 * out fp32: NDHWC [B][V][ldo] (out_ndhwc=1) or NCDHW [B][nsel][V]. */
int dgtta_seghead_fwd(const void *x, int ldx, const float *w, const float *bias, const int *sel, int nsel,
                      float *out, int out_ndhwc, int ldo, int B, int Cin, int64_t V, int dtype, void *stream);
/* dx = dout . W_sel ; dw_sel[nsel][Cin], db_sel[nsel] (+)= (fp32, compact rows; caller scatters). */
size_t dgtta_seghead_bwd_ws_bytes(int B, int Cin, int nsel, int64_t V);
int dgtta_seghead_bwd(const void *x, int ldx, const float *dout, int lddo, const float *w, const int *sel,
                      int nsel, void *dx, int lddx, float *dw_sel, float *db_sel, void *ws, size_t ws_bytes,
                      int B, int Cin, int64_t V, int accumulate, int dtype, void *stream);

/* Segmentation head fused with the inverse warp of its logits: the two steps at the end of calc_branch
 * (dg_tta/tta/tta.py:560-575: model() -> map_label(logits) -> grid_sample(..., R_inverse grid, zeros padding)) in one launch
 * each way.  Both are linear and the head is per voxel, so warp(head(z)) = head(warp(z)) + bias * (in-bounds corner weight):
 * the un-warped logits and their gradient are never materialised.  z: NDHWC [B][D][H][W][32] in `dtype` (bf16 / fp16 only);
 * w / bias: the head's fp32 parameters, sel: the nsel (4, 8, 12 or 16) selected class rows; theta [B][3][4] = R_inverse.
 * fwd: out [B][D][H][W][nsel] fp32 = the branch's logits in the common frame.
 * bwd: gout = gradient of that tensor; gz [B][D][H][W][32] (dtype) = gradient of z; dw_sel [nsel][32], db_sel [nsel] fp32
 * (either may be NULL; accumulate != 0 adds).  h_theta: HOST copy of theta - the owner-computes gather has no scatter
 * fallback here, so maps it would decline (singular / strongly minifying) are rejected with DGTTA_ERR_UNSUPPORTED before
 * anything is launched and the caller uses dgtta_seghead_* + dgtta_affine_warp3d_*.  ws: dgtta_seghead_warp_bwd_ws_bytes
 * (0 = shape not offered: B*D*H*W must be a multiple of 128). */
/* 1 when the fused pair applies to this shape / dtype and to every map of the HOST array h_theta [B][3][4], else 0. */
int dgtta_seghead_warp_supported(const float *h_theta, int B, int Cin, int nsel, int D, int H, int W, int dtype);
size_t dgtta_seghead_warp_bwd_ws_bytes(int B, int Cin, int nsel, int D, int H, int W);
int dgtta_seghead_warp_fwd(const void *z, const float *w, const float *bias, const int *sel, int nsel, const float *theta,
                           float *out, int B, int Cin, int D, int H, int W, int tta_grid_algebra, int dtype, void *stream);
int dgtta_seghead_warp_bwd(const void *z, const float *gout, const float *theta, const float *h_theta, const float *w,
                           const int *sel, int nsel, void *gz, float *dw_sel, float *db_sel, void *ws, size_t ws_bytes, int B,
                           int Cin, int D, int H, int W, int tta_grid_algebra, int accumulate, int dtype, void *stream);
/* The same with gout16 = the gradient of the warped logits in the network's 16-bit storage type `dtype`, rows of nsel values as
 * dgtta_softdice_bwd_t writes them (round 6): the gather reads half the bytes per candidate; acc += float(g16) * weight in the
 * same order, so the result equals dgtta_seghead_warp_bwd on the widened values bit for bit. */
int dgtta_seghead_warp_bwd_g16(const void *z, const void *gout16, const float *theta, const float *h_theta, const float *w,
                               const int *sel, int nsel, void *gz, float *dw_sel, float *db_sel, void *ws, size_t ws_bytes, int B,
                               int Cin, int D, int H, int W, int tta_grid_algebra, int accumulate, int dtype, void *stream);

/* Layout / dtype converters between the reference's NCDHW fp32 and internal NDHWC. */
int dgtta_ncdhw_to_ndhwc(const float *src, void *dst, int B, int C, int64_t V, int ldc, int dtype, void *stream);
int dgtta_ndhwc_to_ncdhw(const void *src, float *dst, int B, int C, int64_t V, int ldc, int dtype, void *stream);

/* argmax over channels + per-label hard Dice counts; replaces tta.py:321 + dice_coeff
 * (torch_utils.py:107-117).  logits NDHWC fp32 (or NULL: predictions are read from argmax_out);
 * labels int64 [B][V] or NULL; argmax_out int64 [B][V]; counts[3*C] int64 (|pred==l|, |gt==l|,
 * |both|), zeroed by the caller. */
int dgtta_argmax_dice(const float *logits, int ldc, int C, const int64_t *labels, int64_t *argmax_out,
                      int64_t *counts, int B, int64_t V, void *stream);

/* Gaussian-weighted sliding-window accumulation for the post-TTA ensemble inference (nnU-Net's
 * predict_sliding_window_return_logits, reached from dg_tta/tta/nnunet_utils.py:116-125,208-230):
 * acc[(x0..,y0..,z0..)][c] += patch[p][c]*gauss[p]; nsum[..] += gauss[p] (nsum may be NULL).
 * patch [PD][PH][PW][C], acc [X][Y][Z][C], nsum [X][Y][Z], all fp32 voxel-major. */
int dgtta_window_accumulate(const float *patch, const float *gauss, float *acc, float *nsum, int C, int PD, int PH,
                            int PW, int X, int Y, int Z, int x0, int y0, int z0, void *stream);

/* The same with the 1x1x1 segmentation head in front of it: acc += gauss * (W z + bias) for ALL C classes of one window,
 * from the window's feature map z [PD][PH][PW][32] (16-bit storage) - the window's logits (880 MB at 128^3 x 105) are never
 * written.  Logits are evaluated as dgtta_seghead_fwd does (identical values). */
int dgtta_seghead_window_accumulate(const void *z, const float *w, const float *bias, const float *gauss, float *acc,
                                    float *nsum, int Cin, int C, int PD, int PH, int PW, int X, int Y, int Z, int x0, int y0,
                                    int z0, int dtype, void *stream);

/* The accumulator's storage type as a parameter (round 4): acc_dtype DGTTA_F32 or DGTTA_F16.  nnU-Net's predictor keeps
 * `predicted_logits` in torch.half (nnunetv2==2.2.1 predict_sliding_window_return_logits); with DGTTA_F16 the sum is formed
 * in fp32 and rounded to half once per window and voxel - half the HBM traffic of the read-modify-write and half the 52.5 GiB
 * of a 105-class 512^3 volume.  nsum stays fp32.  The functions above are these with DGTTA_F32. */
int dgtta_window_accumulate_t(const float *patch, const float *gauss, void *acc, float *nsum, int C, int PD, int PH, int PW,
                              int X, int Y, int Z, int x0, int y0, int z0, int acc_dtype, void *stream);
int dgtta_seghead_window_accumulate_t(const void *z, const float *w, const float *bias, const float *gauss, void *acc,
                                      float *nsum, int Cin, int C, int PD, int PH, int PW, int X, int Y, int Z, int x0, int y0,
                                      int z0, int dtype, int acc_dtype, void *stream);
/* argmax over the C <= 112 classes of `rows` voxel-major rows stored back to back (fp32 or fp16): the label map of the
 * window accumulator (nnU-Net's convert_predicted_logits_to_segmentation..., argmax over ALL pretrain classes,
 * dg_tta/tta/tta.py:404-413).  First maximum wins (dgtta_argmax_dice routes here when it can). */
int dgtta_argmax_rows(const void *logits, int acc_dtype, int C, int64_t rows, int64_t *argmax_out, void *stream);

/* One-axis spline resampling (order 0 / 1 / 3) with the coordinate rule and boundary handling of
 * skimage.transform.resize(mode='edge', anti_aliasing=False) = scipy.ndimage.zoom(mode='nearest', grid_mode=True), which is
 * what nnU-Net's DefaultPreprocessor resamples with (third-party nnunetv2==2.2.1, reached from preprocess_fromfile,
 * dg_tta/tta/nnunet_utils.py:170-204).  src: double [outer][n][inner], dst: double [outer][m][inner]; applied per axis
 * by the caller (the nD operation is separable).  ws: dgtta_resample_axis_ws_bytes (order 3 only). */
size_t dgtta_resample_axis_ws_bytes(int64_t outer, int n, int64_t inner, int order);
int dgtta_resample_axis(const double *src, double *dst, void *ws, size_t ws_bytes, int64_t outer, int n, int m,
                        int64_t inner, int order, void *stream);

/* Export of a prediction in the case's ORIGINAL geometry: nnU-Net's convert_predicted_logits_to_segmentation_with_correct_shape
 * (third-party nnunetv2==2.2.1), which the reference reaches through predict_from_data_iterator
 * (dg_tta/tta/nnunet_utils.py:208-230) before sitk_io.write_seg (dg_tta/tta/tta.py:404-413).
 * dgtta_logits_chunk_f64: dst[x][y][z][j] = acc[x0+x][y0+y][z0+z][c0+j] / nsum[..] as double (classes c0..c0+cg-1 of the
 * window accumulator, cropped back from the padding to the patch size), the input of the dgtta_resample_axis passes.
 * dgtta_argmax_merge_f64: running argmax over class groups: vals double [V][cg] holds classes c0..; best_val / best_idx
 * are initialised when first != 0.  Ties keep the lower class id (argmax semantics). */
int dgtta_logits_chunk_f64(const float *acc, const float *nsum, double *dst, int C, int X, int Y, int Z, int x0, int y0,
                           int z0, int xs, int ys, int zs, int c0, int cg, void *stream);
int dgtta_argmax_merge_f64(const double *vals, int64_t V, int cg, int c0, double *best_val, int *best_idx, int first,
                           void *stream);
int dgtta_logits_chunk_f64_t(const void *acc, const float *nsum, double *dst, int C, int X, int Y, int Z, int x0, int y0,
                             int z0, int xs, int ys, int zs, int c0, int cg, int acc_dtype, void *stream);

/* Sliding-window accumulation in FEATURE space (round 5).  The segmentation head is a 1x1x1 convolution and the last layer of the
 * network, so sum_w g_w (W z_w + b) = W (sum_w g_w z_w) + b sum_w g_w: the volume accumulator holds the 32 Gaussian-weighted feature
 * channels of the last decoder block (128 B per voxel instead of 420 B for 105 fp32 logits; 16 GiB instead of 52.5 GiB at 512^3) and
 * the head runs once per voxel at the end.  Replaces, for the label map the reference writes (dg_tta/tta/tta.py:404-413), the
 * accumulate / average / argmax chain of nnunetv2==2.2.1 predict_sliding_window_return_logits + predict_logits_from_preprocessed_data
 * (reached from dg_tta/tta/nnunet_utils.py:116-125, 208-230); sums are re-associated in fp32 (labels can differ at float ties).
 * dgtta_feature_window_accumulate: facc[(x0.., y0.., z0..)][k] += gauss[p] * z[p][k], nsum[..] += gauss[p] for one window;
 *   z [PD][PH][PW][Cin = 32] in `dtype` storage (DGTTA_F32 / BF16 / F16), facc [X][Y][Z][32] fp32, nsum [X][Y][Z] fp32 or NULL.
 * dgtta_feature_head_argmax: argmax_out[v] = argmax_c sum_m (w[m][c] . facc[m][v]) + nsum[v] * bsum[c] over V voxels; facc member m
 *   starts at facc + m * member_stride (floats), w [M][C][32], bsum [C] = sum over members of the head biases.  First maximum wins.
 * dgtta_feature_logits_chunk_f64: the export path's class chunk (see dgtta_logits_chunk_f64) evaluated from the features in double:
 *   dst[x][y][z][j] = sum_m (w[m][c0 + j] . facc[m][sv]) / nsum[sv] + bsum[c0 + j]. */
int dgtta_feature_window_accumulate(const void *z, const float *gauss, float *facc, float *nsum, int Cin, int PD, int PH, int PW, int X,
                                    int Y, int Z, int x0, int y0, int z0, int dtype, void *stream);
/* ... with the InstanceNorm + LeakyReLU apply of the block in front of the head folded in: y [PD][PH][PW][32] is that block's raw conv
 * output, mean_rstd [32][2] the window's statistics (dgtta_instnorm_lrelu_fwd with z = NULL leaves them without applying them);
 * z = round_dtype(lrelu(y * rstd * gamma + (beta - mean * rstd * gamma))) is formed in registers - the values the apply pass writes. */
int dgtta_feature_window_accumulate_norm(const void *y, const float *mean_rstd, const float *gamma, const float *beta, float slope,
                                         const float *gauss, float *facc, float *nsum, int Cin, int PD, int PH, int PW, int X, int Y,
                                         int Z, int x0, int y0, int z0, int dtype, void *stream);
/* ... for nsrc = 1..4 windows of one network pass that overlap along the last axis, one SEGMENT of that axis per call: h_srcs[k] (a HOST array of device pointers, like h_mean_rstds and h_zoffs) is window
 * k's features (h_mean_rstds == NULL) or raw conv output (h_mean_rstds[k] = its statistics), h_zoffs[k] the segment's first voxel in window
 * k's coordinates, seg_len its length, (x0, y0, z0) its first voxel in the volume.  Contributions are added in window order in
 * registers - the bits nsrc single-window calls produce - and the accumulator is read and written once. */
int dgtta_feature_window_accumulate_multi(const void *const *h_srcs, const float *const *h_mean_rstds, const int *h_zoffs, int nsrc,
                                          const float *gamma, const float *beta, float slope, const float *gauss, float *facc,
                                          float *nsum, int Cin, int PD, int PH, int PW, int seg_len, int X, int Y, int Z, int x0, int y0,
                                          int z0, int dtype, void *stream);
int dgtta_feature_head_argmax(const float *facc, int64_t member_stride, const float *nsum, const float *w, const float *bsum, int M,
                              int Cin, int C, int64_t V, int64_t *argmax_out, void *stream);
int dgtta_feature_logits_chunk_f64(const float *facc, int64_t member_stride, const float *nsum, const float *w, const float *bsum,
                                   double *dst, int M, int Cin, int C, int X, int Y, int Z, int x0, int y0, int z0, int xs, int ys,
                                   int zs, int c0, int cg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DGTTA_H */
