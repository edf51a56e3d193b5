/*
 * mmif.h -- C ABI of the MI355X (gfx950) image-fusion hot-path library (libmmif_hip.so).
 *
 * The reference (chenzpstar/Multi-Modal-Image-Fusion) has no FFI of its own: its hot path is
 * Python calling torch ops.  This header is the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md) to replace those torch calls.  Every entry point below cites the reference
 * call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller; nothing here allocates or frees;
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     immediately: 0 on success, a negative MMIF_E* code on error (message: mmif_last_error());
 *   - no function throws, synchronises the device, or touches the host copy of any tensor, so
 *     all of them are legal inside hipStreamBeginCapture/EndCapture (hipGraph);
 *   - reductions are two-stage and deterministic (no floating-point atomics).
 *
 * Feature-map layout ("blocked NHWC", mmif_tensor): [n][cb_total][hs][ws][8] where one granule
 * = 8 consecutive channels of one pixel (16 B in bf16, 32 B in fp32), hs = h + 2*halo,
 * ws = w + 2*halo.  A view selects channel blocks [cb_off, cb_off+cb): torch.cat(dim=1)
 * (core/fusion.py:38-39) becomes "producers write adjacent channel blocks of one allocation".
 * Activations use halo = 0 (reflect padding is applied by index arithmetic while loading).
 * Gradients w.r.t. activations use halo = 1 ("padded-domain" gradients): a dgrad kernel writes
 * the gradient of the reflect-PADDED input; whoever reads it folds the halo back onto the
 * interior while loading (the adjoint of reflect padding), see DESIGN.md.
 * Single-channel images ([B,1,H,W] in the reference) are plain contiguous fp32 [n][h][w].
 */
#ifndef MMIF_H
#define MMIF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMIF_F32 0
#define MMIF_BF16 1

#define MMIF_IMPL_AUTO 0 /* bf16 -> MFMA kernels; fp32 -> split-bf16 MFMA kernels (X3) for 3x3 layers given their operand image, else VALU */
#define MMIF_IMPL_VALU 1 /* LDS-tiled fp32-accumulate VALU kernels (any dtype) */
#define MMIF_IMPL_MFMA 2 /* v_mfma_f32_16x16x32_bf16 kernels (bf16 storage only) */
#define MMIF_IMPL_X3 3   /* fp32 storage, contraction on the matrix pipe as three bf16 products (hi*hi + hi*lo + lo*hi,
                          * fp32 accumulate): the parity-grade fast path for fp32 tensors, 3x3 and 1x1 layers (csrc/conv_x3.hip) */

#define MMIF_OK 0
#define MMIF_EINVAL (-1)   /* bad argument / unsupported shape */
#define MMIF_ELAUNCH (-2)  /* HIP launch error */
#define MMIF_EWORKSPACE (-3) /* workspace too small */

typedef struct mmif_tensor {
    void* data;       /* base of the allocation (NOT offset to the view) */
    int32_t dtype;    /* MMIF_F32 | MMIF_BF16 */
    int32_t n, h, w;  /* logical extent, halo excluded */
    int32_t halo;     /* 0 (activations) or 1 (padded-domain gradients) */
    int32_t cb_total; /* channel blocks (of 8 channels) in the allocation */
    int32_t cb_off;   /* first channel block of the view */
    int32_t cb;       /* channel blocks in the view */
    int32_t flags;    /* MMIF_T_FOLDED: halo already folded onto the interior and zeroed (mmif_fold_halo) */
} mmif_tensor;

#define MMIF_T_FOLDED 1

const char* mmif_version(void);
const char* mmif_last_error(void);

/* ---- layout helpers (boundary between the reference's NCHW fp32 tensors and the engine) ---- */
/* NCHW fp32 [n][c][h][w] -> view (channels >= c inside the view are zero-filled). */
int mmif_nchw_to_blocked(const float* src, int32_t c, const mmif_tensor* dst, void* stream);
/* view -> NCHW fp32 [n][c][h][w]; for halo=1 views the interior is exported with the halo folded
 * (i.e. the true gradient w.r.t. the unpadded tensor). */
int mmif_blocked_to_nchw(const mmif_tensor* src, float* dst, int32_t c, void* stream);
/* fill a view (all of hs x ws) with zeros */
int mmif_zero(const mmif_tensor* t, void* stream);
/* halo-1 (padded-domain) gradient view: add the halo onto rows/cols 1 and h-2 / w-2 (adjoint of reflect
 * padding; replaces autograd's reflection_pad2d_backward) and zero the halo, in place.  Afterwards the
 * interior IS the gradient w.r.t. the unpadded tensor; readers given MMIF_T_FOLDED skip their fold-on-load.
 * Linear, so it may run after every contribution of an accumulated gradient. */
int mmif_fold_halo(const mmif_tensor* t, void* stream);

/* Pack fp32 master weights [cout][cin][k][k] (nn.Conv2d layout, core/block.py:56-66) into the
 * MFMA operand images: fwd  [k*k][cin/8 ][cout16][8] and dgrad [k*k][cout16/8][cin16][8]
 * (flipped taps, transposed), bf16, cout16/cin16 = rounded up to 16, zero padded. */
size_t mmif_packed_weight_bytes(int32_t cout, int32_t cin, int32_t ksize);
int mmif_pack_weights(const float* w, int32_t cout, int32_t cin, int32_t ksize, void* packed_fwd,
                      void* packed_dgrad, void* stream);
/* The same for every layer of a model in one launch (host array of jobs; either image pointer may be NULL).
 * What a training step calls after the optimiser has changed the master weights. */
#define MMIF_PACK_BF16 0 /* mmif_pack_weights' images (bf16 tensors) */
#define MMIF_PACK_X3 1   /* mmif_pack_weights_x3's images (fp32 tensors, split-bf16 kernels) */
typedef struct mmif_pack_job {
    const float* w;
    int32_t cout, cin, ksize;
    int32_t format;   /* MMIF_PACK_BF16 | MMIF_PACK_X3 */
    void* packed_fwd;
    void* packed_dgrad;
} mmif_pack_job;
int mmif_pack_weights_multi(const mmif_pack_job* jobs, int32_t n_jobs, void* stream);
/* Operand images of the split-bf16 ("x3") kernels that run the 3x3 ConvLayers of FP32 tensors on the matrix pipe at fp32-grade
 * accuracy (core/block.py:56-66 computes in fp32; BASELINE north star: within 1e-3 of it): every weight is stored as successive bf16
 * pieces hi = bf16(w), mid = bf16(w - hi) [, lo = bf16(w - hi - mid)]; layout [m-block][16-channel chunk][piece][tap][2 channel
 * blocks][32 or 64 out channels][8] (csrc/conv_x3.hip).
 * Pass them as w_packed / w_packed_t of the conv entry points below when the tensors are fp32.  ksize 1 or 3. */
/* Operand format of the FORWARD image / kernels: 16 (default) = two SCALED FP16 pieces (2^10 w = hi + lo), three products per tap,
 * 2^-23 per product -- fp32-grade activations, so ReLU decisions agree with the reference's as often as between two fp32
 * implementations; activations are scaled per staged tile (any magnitude up to 2^114), weights saturate at |w| >= 64;  3 = three bf16
 * pieces, six products, the same accuracy on bf16's own range at twice the forward MFMA work;  2 = two bf16 pieces, three products, activations within
 * ~1e-5 (a mask flip on a pre-activation that close to zero moves parameter gradients by O(1e-3)).  The dgrad image always has two bf16
 * pieces (the backward kernels are linear in the gradient: no decisions to protect).  Process-wide setting, also
 * $MMIF_X3_FWD_PIECES.  It selects the format of the NEXT packs: the library remembers the format each forward image was packed in
 * (by device address) and a forward launch reads the image in THAT format, whatever the setting is by then (round 4; an image packed by
 * another process / library instance is read in the current setting). */
void mmif_set_x3_forward_pieces(int32_t pieces);
int32_t mmif_get_x3_forward_pieces(void);
/* 1 when fp32 tensors are taken by the split-operand kernels ($MMIF_X3 != 0, read once per process); what a caller that plans launches
 * around them (sign-map backward, fused encoder weight gradients) must consult instead of the environment. */
int32_t mmif_get_x3_enabled(void);
/* How many weight values the scaled-fp16 forward operand images (forward pieces = 16: the default) CLAMPED since the last reset: the fixed 2^10
 * weight scale holds |w| < ~63.5.  Synchronises the device: call after loading / initialising weights, never per step.  > 0 means the
 * fp32 forward is not parity-grade for those layers -- use mmif_set_x3_forward_pieces(3).  reset != 0 clears the counter.  -1 on error. */
int32_t mmif_x3_pack_saturations(int32_t reset);
size_t mmif_packed_weight_bytes_x3(int32_t cout, int32_t cin, int32_t ksize);
int mmif_pack_weights_x3(const float* w, int32_t cout, int32_t cin, int32_t ksize, void* packed_fwd, void* packed_dgrad, void* stream);

/* ---- General ConvLayer primitives (row n4: the nets outside the PFNet/DenseFuse hot path) on plain NCHW fp32 tensors ----
 * nn.Conv2d(cin, cout, k in {1,3,5,7}, stride in {1,2}, padding <= k/2, padding_mode reflect|zeros) (+ ReLU)
 * replaces core/block.py:56-66 for DeepFuse (core/model.py:152-158, k = 5/7), DBNet (:219-221, stride 2), NestFuse/UNFusion
 * down_mode='stride' (:338-340).  x [n][cin][h][w], w [cout][cin][k][k], y [n][cout][ho][wo], ho = (h + 2p - k)/s + 1. */
int mmif_gconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t cin, int32_t cout,
                   int32_t h, int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, int32_t relu,
                   void* stream);
/* dx (h x w) from gy (ho x wo, already multiplied by the activation's derivative); reflect padding needs a workspace of
 * mmif_gconv_dgrad_workspace bytes (the padded-domain map that the reflect adjoint folds). */
size_t mmif_gconv_dgrad_workspace(int32_t n, int32_t cin, int32_t h, int32_t wd, int32_t padding, int32_t reflect);
int mmif_gconv_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t wd,
                     int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, void* workspace, size_t workspace_bytes,
                     void* stream);
/* dw [cout][cin][k][k], db [cout] (may be NULL); deterministic two-stage sum. */
size_t mmif_gconv_wgrad_workspace(int32_t cin, int32_t cout, int32_t ksize);
int mmif_gconv_wgrad(const float* x, const float* gy, float* dw, float* db, int32_t n, int32_t cin, int32_t cout, int32_t h,
                     int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t reflect, void* workspace,
                     size_t workspace_bytes, void* stream);
/* nn.ConvTranspose2d(cin, cout, k, stride, padding, output_padding) (core/block.py:67-76; SEDRFuse core/model.py:258-259).
 * x [n][cin][h][w], w [cin][cout][k][k], y [n][cout][ho][wo], ho = (h-1) s - 2p + k + output_padding.  wgrad: db_scratch receives
 * per-channel sums of x (NOT a layer gradient; pass NULL) -- the bias gradient is the plane sum of gy. */
int mmif_gconvt_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t cin, int32_t cout,
                    int32_t h, int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding, int32_t relu,
                    void* stream);
int mmif_gconvt_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t wd,
                      int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding, void* stream);
int mmif_gconvt_wgrad(const float* x, const float* gy, float* dw, float* db_scratch, int32_t n, int32_t cin, int32_t cout,
                      int32_t h, int32_t wd, int32_t ksize, int32_t stride, int32_t padding, int32_t output_padding,
                      void* workspace, size_t workspace_bytes, void* stream);
/* depth-wise ConvLayer (groups == channels, Res2ConvBlock.dwconvs core/block.py:317-325): w [c][1][k][k], k in {1,3}, stride 1, padding k/2 */
int mmif_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t c, int32_t h, int32_t wd,
                    int32_t ksize, int32_t reflect, void* stream);
int mmif_dwconv_dgrad(const float* gy, const float* w, float* dx, int32_t n, int32_t c, int32_t h, int32_t wd, int32_t ksize,
                      int32_t reflect, void* stream);
int mmif_dwconv_wgrad(const float* x, const float* gy, float* dw, float* db, int32_t n, int32_t c, int32_t h, int32_t wd,
                      int32_t ksize, int32_t reflect, void* stream);
/* out = g * [y > 0] on plain fp32 arrays (ReLU backward of the layers above) */
int mmif_relu_bwd(const float* g, const float* y, float* out, int64_t count, void* stream);
/* out[c] = sum_{n, pixels} x[n][c][.] (deterministic; the bias gradient of a ConvTranspose2d) */
int mmif_channel_sum(const float* x, float* out, int32_t n, int32_t c, int64_t hw, void* stream);

/* nn.Upsample(scale_factor, mode='bilinear', align_corners=True) (core/block.py:965-973; DBNet core/model.py:223, the up_mode option
 * of NestFuse / UNFusion / MAFusion) on `planes` = n * c planes of plain fp32 [h][w] -> [H][W]; backward = its adjoint, gathered. */
int mmif_bilinear_up_fwd(const float* x, float* out, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, void* stream);
int mmif_bilinear_up_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t H, int32_t W, void* stream);

/* Norm + activation epilogue of ConvLayer (core/block.py:78-92) on plain NCHW fp32: y = act(gamma * (x - mean) * rstd + beta).
 * kind 0 = nn.BatchNorm2d in training mode (batch statistics; running_mean / running_var, when given, are updated with `momentum`
 * and the unbiased variance), 1 = nn.BatchNorm2d in eval mode (running buffers), 2 = nn.GroupNorm(c, c) (one group per channel:
 * per-(sample, channel) statistics; SEDRFuse core/model.py:249-260).  act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 Tanh, 4 ReLU6.
 * stats receives (mean, rstd) per channel (kinds 0, 1: 2 c floats) or per plane (kind 2: 2 n c floats) for the backward pass.
 * Backward: dx, dgamma, dbeta (either may be NULL) from x, y (the forward's output), gy. */
size_t mmif_norm_workspace(int32_t n, int32_t c);
int mmif_norm_act_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* running_mean,
                      float* running_var, int32_t n, int32_t c, int64_t hw, int32_t kind, float eps, float momentum, int32_t act,
                      float slope, void* workspace, size_t workspace_bytes, void* stream);
int mmif_norm_act_bwd(const float* x, const float* y, const float* gy, const float* stats, const float* gamma, float* dx,
                      float* dgamma, float* dbeta, int32_t n, int32_t c, int64_t hw, int32_t kind, int32_t act, float slope,
                      void* workspace, size_t workspace_bytes, void* stream);
/* Cross-rank form of kind 0 (the reference converts its BatchNorm nets with nn.SyncBatchNorm.convert_sync_batchnorm before DDP,
 * train.py:296): statistics and apply stages are separate calls; between them the host all-reduces (SUM) the fp64 device buffer
 * chan_sums.  Forward: [2c + 1] doubles = (sum x, sum x^2) per channel, then the element count n*hw, so the count is reduced by the
 * same collective and never visits the host.  Backward: [2c] doubles = (sum dz, sum dz*xhat) per channel; `count` points at the
 * forward's reduced count (device memory).  dgamma / dbeta of mmif_bn_bwd_sums are the rank-local sums (they ride in the gradient
 * all-reduce, as torch's SyncBatchNorm leaves them to DDP).  Without an all-reduce the results equal kind 0's. */
int mmif_bn_moments(const float* x, double* chan_sums, int32_t n, int32_t c, int64_t hw, void* workspace, size_t workspace_bytes,
                    void* stream);
int mmif_bn_apply_fwd(const float* x, const double* chan_sums, const float* gamma, const float* beta, float* y,
                      float* stats, float* running_mean, float* running_var, int32_t n, int32_t c, int64_t hw, float eps, float momentum,
                      int32_t act, float slope, void* stream);
int mmif_bn_bwd_sums(const float* x, const float* y, const float* gy, const float* stats, double* chan_sums, float* dgamma, float* dbeta,
                     int32_t n, int32_t c, int64_t hw, int32_t act, float slope, void* workspace, size_t workspace_bytes, void* stream);
int mmif_bn_apply_bwd(const float* x, const float* y, const float* gy, const float* stats, const float* gamma, const double* chan_sums,
                      const double* count, float* dx, int32_t n, int32_t c, int64_t hw, int32_t act, float slope, void* stream);
/* activation alone (LeakyReLU / Tanh after a conv without norm: PMGI's decode, core/model.py:579); backward from the output y */
int mmif_act_fwd(const float* x, float* y, int64_t count, int32_t act, float slope, void* stream);
int mmif_act_bwd(const float* gy, const float* y, float* dx, int64_t count, int32_t act, float slope, void* stream);

/* Resampling glue of the layer-by-layer blocks on plain NCHW fp32 planes: nn.MaxPool2d(k, k) (floor mode; idx = window offset of the
 * first maximum, 1 byte per output; core/block.py:941-950), nn.Upsample(scale_factor, mode='nearest') (:968-969), and
 * nn.ReflectionPad2d((left, right, top, bottom)) as Upsample / Downsample use it to match a target shape (:983-991; negative = crop). */
int mmif_maxpool_nchw_fwd(const float* x, float* y, unsigned char* idx, int64_t planes, int32_t h, int32_t w, int32_t k, void* stream);
int mmif_maxpool_nchw_bwd(const float* g, const unsigned char* idx, float* dx, int64_t planes, int32_t h, int32_t w, int32_t k, void* stream);
int mmif_nearest_up_fwd(const float* x, float* y, int64_t planes, int32_t h, int32_t w, int32_t scale, void* stream);
int mmif_nearest_up_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t scale, void* stream);
int mmif_reflect_pad_fwd(const float* x, float* y, int64_t planes, int32_t h, int32_t w, int32_t left, int32_t right, int32_t top,
                         int32_t bottom, void* stream);
int mmif_reflect_pad_bwd(const float* g, float* dx, int64_t planes, int32_t h, int32_t w, int32_t left, int32_t right, int32_t top,
                         int32_t bottom, void* stream);

/* ---- ConvLayer: reflect-pad(k/2) conv + bias + ReLU, stride 1, k in {1,3}
 *      replaces core/block.py:98-99 (nn.Conv2d(padding_mode='reflect') + nn.ReLU(inplace)) ---- */
/* y = act(bias + corr(reflect_pad(x), w)).  w: fp32 master weights; w_packed: mmif_pack_weights'
 * fwd image for bf16 tensors, mmif_pack_weights_x3's for fp32 tensors (NULL: the VALU kernels). */
int mmif_conv2d_reflect_fwd(const mmif_tensor* x, const float* w, const void* w_packed, const float* bias,
                            const mmif_tensor* y, int32_t cin, int32_t cout, int32_t ksize, int32_t relu,
                            int32_t impl, void* stream);
/* gx(+)= full-correlation(fold(gy), w^T) on the padded domain [h+2p][w+2p].
 * gy: gradient w.r.t. the conv's (post-activation) output, halo 0 or 1, ALREADY masked by the
 *     layer's own ReLU.  gx: halo = 1 view.
 * mask_bits / accum_bits: bit i refers to channel block i of the gx view:
 *   accum: gx = gx_old + new (another consumer of x already wrote its contribution);
 *   mask : multiply by [x > 0] (x = this conv's forward input = previous layer's ReLU output),
 *          applied after accumulation; x may be NULL when mask_bits == 0.
 * replaces autograd's convolution_backward(input) + reflection_pad2d_backward + threshold_backward. */
int mmif_conv2d_reflect_dgrad(const mmif_tensor* gy, const float* w, const void* w_packed_t, const mmif_tensor* x,
                              const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits,
                              uint64_t accum_bits, int32_t impl, void* stream);
/* The same followed by mmif_fold_halo(gx): on return gx's interior is the gradient w.r.t. the unpadded tensor and its halo ring
 * is zero (treat it as MMIF_T_FOLDED).  gx's halo ring must be zero on entry (a fresh zeroed buffer, or the result of an
 * earlier fold), also when accumulating.  The bf16 DMA-staged kernels and (round 6) the 32-wide-tile split-operand kernels of fp32 tensors
 * do the fold inside the border tiles of the dgrad (interior tiles only, no second pass, the halo values are never rounded); every other
 * case runs dgrad + the fold kernel. */
int mmif_conv2d_reflect_dgrad_folded(const mmif_tensor* gy, const float* w, const void* w_packed_t, const mmif_tensor* x,
                                     const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize,
                                     uint64_t mask_bits, uint64_t accum_bits, int32_t impl, void* stream);
/* The same with the accumulate operand taken from ANOTHER tensor: gx = [mask] (fold(dgrad(gy)) + gx_old) on the blocks in accum_bits.
 * gx_old: gx's shape / halo / channel blocks, folded.  bf16 thin layers only (the asynchronous kernel): ask
 * mmif_conv2d_dgrad_onto_supported first.  Use: DenseFuse / VIFNet, where both encoder branches start from the ONE gradient of
 * f1 + f2 (core/fusion.py:21-29 'sum') -- no per-branch copy of it. */
/* Folded dgrad of a wide 3x3 layer (its own output neither masked nor accumulated) that ALSO leaves two masked copies of fragment `frag`
 * (16 channels = channel blocks 2 frag, 2 frag + 1 of gx) in dup_out (round 6, DenseFuse `core/model.py:165-186`: the gradient of f1 + f2 is
 * the gradient of either encoder's output, but each encoder's last DenseBlock conv still owes it the ReLU mask of its own x3):
 *   dup_out blocks [2 frag, + 1]     = gx blocks [2 frag, + 1] * [dup_mask blocks [2 frag, + 1]     > 0]
 *   dup_out blocks [2 frag + 8, + 1] = gx blocks [2 frag, + 1] * [dup_mask blocks [2 frag + 8, + 1] > 0]
 * dup_out: bf16 halo-1 tensor of gx's shape (interior written, ring untouched: folded), dup_mask: the bf16 halo-0 activations.  Bit-identical
 * to mmif_conv2d_reflect_dgrad_folded + mmif_fuse_elem_bwd(MMIF_FUSE_SUM, relu_mask) on those blocks. */
int mmif_conv2d_dgrad_dup_supported(const mmif_tensor* gy, const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize);
int mmif_conv2d_reflect_dgrad_folded_dup(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* gx, int32_t cin, int32_t cout,
                                         int32_t ksize, const mmif_tensor* dup_out, const mmif_tensor* dup_mask, int32_t frag, void* stream);
int mmif_conv2d_dgrad_onto_supported(const mmif_tensor* gy, const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize);
int mmif_conv2d_reflect_dgrad_folded_onto(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx_old,
                                          const mmif_tensor* gx, int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits,
                                          uint64_t accum_bits, void* stream);
/* dw[cout][cin][k][k] (=|+=) sum_p fold(gy)[p] * reflect_pad(x)[p+tap]; db[cout] (=|+=) sum_p fold(gy)[p].
 * replaces convolution_backward(weight, bias). */
size_t mmif_conv2d_wgrad_workspace(int32_t cin, int32_t cout, int32_t ksize);
int mmif_conv2d_reflect_wgrad(const mmif_tensor* x, const mmif_tensor* gy, float* dw, float* db, int32_t cin,
                              int32_t cout, int32_t ksize, int32_t accumulate, void* workspace,
                              size_t workspace_bytes, int32_t impl, void* stream);

/* ---- image-side layers: Cin == 1 (first encoder conv, core/model.py:73,77,118,169,326) and
 *      Cout == 1 (last decoder conv, core/model.py:86,131,179,344).  Images are fp32 [n][h][w]. */
int mmif_conv2d_image_in_fwd(const float* img, const float* w, const float* bias, const mmif_tensor* y,
                             int32_t cout, int32_t ksize, int32_t relu, void* stream);
int mmif_conv2d_image_in_wgrad(const float* img, const mmif_tensor* gy, float* dw, float* db, int32_t cout,
                               int32_t ksize, int32_t accumulate, void* workspace, size_t workspace_bytes,
                               void* stream);
int mmif_conv2d_image_out_fwd(const mmif_tensor* x, const float* w, const float* bias, float* img, int32_t cin,
                              int32_t ksize, int32_t relu, void* stream);
/* gimg: dL/d(output image) [n][h][w]; y_img: the layer's output when it has a ReLU (else NULL). */
int mmif_conv2d_image_out_dgrad(const float* gimg, const float* y_img, const float* w, const mmif_tensor* x,
                                const mmif_tensor* gx, int32_t cin, int32_t ksize, uint64_t mask_bits,
                                uint64_t accum_bits, void* stream);
int mmif_conv2d_image_out_wgrad(const mmif_tensor* x, const float* gimg, const float* y_img, float* dw, float* db,
                                int32_t cin, int32_t ksize, int32_t accumulate, void* workspace,
                                size_t workspace_bytes, void* stream);

/* The whole backward of the decoders' last layer -- ConvLayer(16, 1, 3x3) of PFNetv1 / DenseFuse / PFNetv2 / VIFNet, reference
 * core/model.py:86,178 -- as ONE launch (round 6, csrc/image_bwd.hip): dL/dx with the reflect-padding adjoint applied and the ReLU mask of
 * the layer's input (x > 0), dW and db.  Replaces mmif_conv2d_image_out_wgrad + mmif_conv2d_image_out_dgrad + mmif_fold_halo on that layer.
 *   x      the layer's input activations, 16 channels (2 channel blocks), halo 0, bf16
 *   gimg   dL/dy, [n][h][w] fp32;  y_img: the layer's output when it has a ReLU (the gradient is masked by y > 0), else NULL
 *   w      fp32 [1][16][3][3]
 *   gx     dL/dx, 2 channel blocks, halo 1: the interior is written (every channel block masked by x > 0, one rounding); the halo ring is
 *          NOT touched and must be zero on entry -- the result is a FOLDED gradient (MMIF_T_FOLDED semantics)
 *   dw, db as mmif_conv2d_image_out_wgrad (workspace: mmif_conv2d_image_wgrad_workspace(16, 3))
 * mmif_conv2d_image_out_bwd_supported says whether (dtype, cin, ksize, h, w) is taken: bf16, 16 channels, 3x3, h, w >= 4; otherwise the
 * call returns MMIF_EINVAL and the caller uses the three separate entry points. */
int32_t mmif_conv2d_image_out_bwd_supported(int32_t dtype, int32_t cin, int32_t ksize, int32_t h, int32_t w);
int mmif_conv2d_image_out_bwd(const mmif_tensor* x, const float* gimg, const float* y_img, const float* w, const mmif_tensor* gx,
                              float* dw, float* db, int32_t cin, int32_t ksize, int32_t accumulate, void* workspace,
                              size_t workspace_bytes, void* stream);
size_t mmif_conv2d_image_wgrad_workspace(int32_t c, int32_t ksize);

/* ---- fusion functions (core/fusion.py) ---- */
#define MMIF_FUSE_SUM 0
#define MMIF_FUSE_MEAN 1
#define MMIF_FUSE_MAX 2
/* element_fusion core/fusion.py:21-29; a, b, out: halo-0 views of equal shape */
int mmif_fuse_elem_fwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* out, int32_t mode,
                       void* stream);
/* g: halo 0 or 1 (folded on load); ga/gb: halo-0 views.  mask_a/mask_b: multiply by [a>0]/[b>0]
 * (when a, b are ReLU outputs whose only consumer is this fusion). */
int mmif_fuse_elem_bwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                       const mmif_tensor* gb, int32_t mode, int32_t relu_mask, void* stream);

/* attention_fusion core/fusion.py:42-59 with spatial 'l1' (:89-90) / channel 'avg' (:123-124) pooling, softmax=False;
 * mode 0 'sa', 1 'ca', 2 'sca'.  a, b, out: halo-0 views.  bwd: g halo 0/1; ga/gb gradient views (interior written,
 * optionally accumulated); gradients flow through the pooled weights as in the reference. */
size_t mmif_fuse_attn_workspace(int32_t n, int32_t c);
int mmif_fuse_attn_fwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* out, int32_t mode, void* workspace,
                       size_t workspace_bytes, void* stream);
int mmif_fuse_attn_bwd(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga,
                       const mmif_tensor* gb, int32_t mode, int32_t accumulate, void* workspace, size_t workspace_bytes,
                       void* stream);
/* ... when `workspace` still holds what mmif_fuse_attn_fwd left there for the same a, b, mode: the channel sums are reused */
int mmif_fuse_attn_bwd_cached(const mmif_tensor* a, const mmif_tensor* b, const mmif_tensor* g, const mmif_tensor* ga, const mmif_tensor* gb,
                              int32_t mode, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- PFNetv2's self-learned fusion (core/model.py:120-124,134-141): the conv stack ConvLayer(2,2) -> ConvLayer(2,2) ->
 *      ConvLayer(2,1,act=None) applied to every channel pair (feat1[:,i], feat2[:,i]) with SHARED weights.  One "pair conv"
 *      launch replaces the 64 per-channel nn.Conv2d calls of one layer of that Python loop: every channel c of the two
 *      operand views a, b is an independent 2-channel image,
 *          oa[c] = act(bias[0] + corr(reflect_pad(a[c]), w[0][0]) + corr(reflect_pad(b[c]), w[0][1]))   (ob: w[1], nout = 2)
 *      w: fp32 [nout][2][3][3] (nn.Conv2d layout), bias fp32 [nout] or NULL.  res1/res2 (both or neither): added to oa
 *      (the "+ feat1 + feat2" of core/model.py:141). ---- */
int mmif_pairconv_fwd(const mmif_tensor* a, const mmif_tensor* b, const float* w, const float* bias, int32_t nout,
                      const mmif_tensor* oa, const mmif_tensor* ob, int32_t relu, const mmif_tensor* res1,
                      const mmif_tensor* res2, void* stream);
/* gxa/gxb (halo-1 views): padded-domain gradient w.r.t. a, b from ga (and gb when nout = 2; already masked by the layer's
 * own ReLU); `add` (optional, halo 0/1): gradient of a residual path, added to both; mask_bits: channel blocks multiplied
 * by [xa > 0] / [xb > 0].  The caller folds the halo afterwards (mmif_fold_halo). */
int mmif_pairconv_dgrad(const mmif_tensor* ga, const mmif_tensor* gb, const float* w, int32_t nout, const mmif_tensor* xa,
                        const mmif_tensor* xb, const mmif_tensor* gxa, const mmif_tensor* gxb, uint64_t mask_bits,
                        const mmif_tensor* add, void* stream);
/* dw[nout][2][3][3] (=|+=), db[nout] (=|+=): summed over batch, channels and pixels (deterministic two-stage). */
size_t mmif_pairconv_wgrad_workspace(void);
int mmif_pairconv_wgrad(const mmif_tensor* xa, const mmif_tensor* xb, const mmif_tensor* ga, const mmif_tensor* gb,
                        int32_t nout, float* dw, float* db, int32_t accumulate, void* workspace, size_t workspace_bytes,
                        void* stream);
/* mmif_pairconv_dgrad + mmif_pairconv_wgrad of one layer in ONE pass (they read the same two tensors: ga/gb and xa/xb):
 * gxa/gxb bit-identical to mmif_pairconv_dgrad, dw/db equal to mmif_pairconv_wgrad up to fp32 summation order.  xa/xb are
 * always required (the weight gradient's operand); the workspace is mmif_pairconv_wgrad_workspace() bytes. */
int mmif_pairconv_bwd(const mmif_tensor* ga, const mmif_tensor* gb, const float* w, int32_t nout, const mmif_tensor* xa,
                      const mmif_tensor* xb, const mmif_tensor* gxa, const mmif_tensor* gxb, uint64_t mask_bits,
                      const mmif_tensor* add, float* dw, float* db, int32_t accumulate, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- NestFuse glue: nn.MaxPool2d(2,2) (core/model.py:332-335), nn.Upsample(x2,'nearest') + ReflectionPad2d to the
 *      skip's shape (core/block.py:965-991), threshold_backward of a ReLU output ---- */
int mmif_maxpool2x2_fwd(const mmif_tensor* x, const mmif_tensor* y, void* stream);
int mmif_maxpool2x2_bwd(const mmif_tensor* x, const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream);
int mmif_upsample2x_fwd(const mmif_tensor* x, const mmif_tensor* y, void* stream);
int mmif_upsample2x_bwd(const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream);
int mmif_relu_mask(const mmif_tensor* x, const mmif_tensor* g, void* stream); /* g *= [x > 0], in place */
/* the two backward kernels with that threshold_backward applied to what they write (the LAST contribution to the gradient of a ReLU
 * output: the pool's own input x / the given x) -- saves the separate pass over the gradient */
int mmif_maxpool2x2_bwd_relu(const mmif_tensor* x, const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, void* stream);
int mmif_upsample2x_bwd_relu(const mmif_tensor* g, const mmif_tensor* gx, int32_t accumulate, const mmif_tensor* x, void* stream);

/* ---- losses (core/loss.py); images fp32 [n][h][w] ---- */
size_t mmif_loss_workspace(int32_t n, int32_t h, int32_t w);
/* SSIMLoss(mode='ssim') core/loss.py:252-257,284 = weight*(1 - (mean SSIM(img1,f)+mean SSIM(img2,f))/2),
 * 11x11 Gaussian window sigma 1.5, valid correlation, data_range given.
 * loss_out: device float[1]; grad_out (may be NULL): dloss/df [n][h][w]. */
int mmif_ssim_loss(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w,
                   float weight, float data_range, float* loss_out, float* grad_out, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The other SSIMLoss modes core/loss.py:259-277: mode 1 'w-ssim' (per-sample sigma weights), 2 'ms-ssim' (5-level
 * avg-pool pyramid, :113-160; images >= 161x161), 3 'msw-ssim' (windows 11/9/7/5/3, per-pixel sigma weights, :211-237).
 * Same contract as mmif_ssim_loss (mode 0 'ssim' stays there); any other mode -> MMIF_EINVAL with the reference's message. */
size_t mmif_ssim_loss_mode_workspace(int32_t n, int32_t h, int32_t w, int32_t mode);
int mmif_ssim_loss_mode(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w, float weight,
                        float data_range, int32_t mode, float* loss_out, float* grad_out, void* workspace, size_t workspace_bytes,
                        void* stream);
/* SSIM module core/loss.py:163-185 (calc_ssim :52-110, size_average=True): per-sample means of the ssim map, the cs map and the
 * clamped source variance of (img1, img2), window 3 | 5 | 7 | 9 | 11 -> out[3][n] on the device.  Values only; workspace as
 * mmif_ssim_loss_mode_workspace(n, h, w, 1). */
int mmif_ssim_terms(const float* img1, const float* img2, int32_t n, int32_t h, int32_t w, int32_t win_size, float data_range,
                    float* out, void* workspace, size_t workspace_bytes, void* stream);
/* TVLoss core/loss.py:347-358 = NormLoss(l1|l2)(x[1:] - x[:-1]) + NormLoss(x[:, 1:] - x[:, :-1]) over n images [h][w]. */
size_t mmif_tv_loss_workspace(void);
int mmif_tv_loss(const float* x, int32_t n, int32_t h, int32_t w, float weight, int32_t l2, float* loss_out, float* grad_out,
                 void* workspace, size_t workspace_bytes, void* stream);
/* PixelLoss core/loss.py:287-304 (NormLoss 'l1'|'l2' :361-385); mode 0 = 'avg', 1 = 'max'. */
int mmif_pixel_loss(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w,
                    float weight, int32_t mode_max, int32_t l2, float* loss_out, float* grad_out, void* workspace,
                    size_t workspace_bytes, void* stream);
/* GradLoss core/loss.py:307-344: Sobel |gx|+|gy| on the reflect-padded image. */
int mmif_grad_loss(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w,
                   float weight, int32_t mode_max, int32_t l2, float* loss_out, float* grad_out, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The three loss terms of the reference's train step (train.py:64-69: SSIMLoss('ssim') + PixelLoss + GradLoss, their sum, the three
 * d/dimgf contributions autograd adds up) as ONE call: the kernels of the three entry points above, the pixel and Sobel kernels adding
 * onto the SSIM term's gradient, one finish kernel.  loss_out = {l1 + l2 + l3, l1, l2, l3, total again} (5 floats on the device); grad_out (or NULL)
 * = d(total)/d(imgf).  Weights as the modules' `weight`; *_max = mode 'max' (else 'avg'); *_l2 = mode 'l2' (else 'l1'). */
size_t mmif_fusion_loss_workspace(int32_t n, int32_t h, int32_t w);
int mmif_fusion_loss(const float* img1, const float* img2, const float* imgf, int32_t n, int32_t h, int32_t w, float w_ssim,
                     float data_range, float w_pixel, int32_t pixel_max, int32_t pixel_l2, float w_grad, int32_t grad_max,
                     int32_t grad_l2, float* loss_out, float* grad_out, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of ONE thin 3x3 ConvLayer (64 -> 32, 32 -> 16: decode.2 / decode.3 of every PFNet / DenseFuse decoder, core/model.py:83-85) in one
 * launch: gx = [x > 0] * dgrad(gy) in the folded convention -- what mmif_conv2d_reflect_dgrad_folded(mask_bits = all, accum_bits = 0)
 * writes, bit for bit -- AND dw / db as mmif_conv2d_reflect_wgrad, from a single staging of the gradient and activation tiles (160 / 80
 * channel planes instead of 256 / 128).  gy: folded halo-1; x: the layer's forward input (halo 0); gx: halo 1 with a zero ring.
 * Workspace: mmif_conv2d_wgrad_workspace(cin, cout, 3).  mmif_conv2d_bwd_pair_supported tells which layers it covers. */
int mmif_conv2d_bwd_pair_supported(int32_t cin, int32_t cout, int32_t ksize);
int mmif_conv2d_reflect_bwd_pair(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx, float* dw, float* db,
                                 int32_t cin, int32_t cout, int32_t ksize, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* Backward of ONE wide 3x3 ConvLayer (Cin, Cout multiples of 64: decode.0 128 -> 128 and decode.1 128 -> 64 of PFNetv1, core/model.py:83-84;
 * DenseFuse's decode.0) as one call = mmif_conv2d_reflect_wgrad followed by mmif_conv2d_reflect_dgrad_folded(mask_bits, accum_bits = 0), with
 * one difference in HBM traffic: the weight-gradient kernel, which stages the activation tiles anyway, leaves their ReLU SIGN BYTES (one
 * byte per pixel and 8 channels) in `signs`, and the input-gradient kernel reads those instead of the activations (1/16 of the bytes).
 * gy: folded halo-1; x: the layer's forward input (halo 0); gx: halo 1 with a zero ring; mask_bits: channel blocks of gx masked with
 * [x > 0].  Workspace: mmif_conv2d_wgrad_workspace(cin, cout, 3); signs: mmif_conv2d_bwd_wide_signs_bytes(n, cin, h, w) bytes, scratch.
 * fp32 tensors (mmif_conv2d_bwd_wide_supported_f32: any 3x3 / 1x1 layer the split-operand kernels of csrc/conv_x3.hip take, x3 operand
 * image as w_packed_t): the same contract -- the split-operand weight gradient leaves the map ([n][ceil(cb / 4)][h][w] dwords, one byte per
 * pixel and channel block), the split-operand dgrad masks with it: 1/32 of the bytes of x.
 * `accumulate`: bit 0 = add onto dw / db; bits 1-2 = phase -- 0: both halves; 1 (value 2): weight gradient + sign map only; 2 (value 4):
 * input gradient only, reading the map a phase-1 call left.  Two calls (2 | a, then 4) give the bits of one call with phase 0 and let a
 * caller time the two kernels apart (bench.py's per-kernel roofline). */
int mmif_conv2d_bwd_wide_supported(int32_t cin, int32_t cout, int32_t ksize);
int mmif_conv2d_bwd_wide_supported_f32(int32_t cin, int32_t cout, int32_t ksize);
size_t mmif_conv2d_bwd_wide_signs_bytes(int32_t n, int32_t cin, int32_t h, int32_t w);
int mmif_conv2d_reflect_bwd_wide(const mmif_tensor* gy, const void* w_packed_t, const mmif_tensor* x, const mmif_tensor* gx, float* dw, float* db,
                                 int32_t cin, int32_t cout, int32_t ksize, uint64_t mask_bits, int32_t accumulate, void* workspace,
                                 size_t workspace_bytes, void* signs, size_t signs_bytes, void* stream);

/* ---- streaming DenseBlock encoder (core/model.py:73-80 = ConvLayer(1,16) + DenseBlock(16,16): PFNetv1.encode1/2, DenseFuse / PFNetv2 /
 *      VIFNet .encode), forward, bf16: the four layers as ONE line-buffer kernel (csrc/enc_stream.hip) -- reads the image, writes the
 *      64 concatenated channels [x0 | x1 | x2 | x3] once.  Bit-identical to mmif_conv2d_image_in_fwd + three mmif_conv2d_reflect_fwd
 *      (MFMA) calls on the same operands.  `packed[i]` = forward operand image (mmif_pack_weights) of DenseBlock conv i (16+16i -> 16,
 *      k = 3); every layer has bias (NULL = zeros) and ReLU.  enc_b / out_b: a second, independent branch in the same launch (other
 *      image, other or the same weights) or NULL.  out_*: bf16 halo-0 views of 8 channel blocks. */
typedef struct mmif_dense_encoder {
    const float* img;         /* [n][h][w] fp32 */
    const float* w0;          /* [16][1][3][3] fp32 */
    const float* b0;          /* [16] or NULL */
    const void* packed[3];
    const float* bias[3];     /* [16] each, or NULL */
} mmif_dense_encoder;
int mmif_dense_encoder_fwd(const mmif_dense_encoder* enc_a, const mmif_tensor* out_a, const mmif_dense_encoder* enc_b,
                           const mmif_tensor* out_b, void* stream);
/* DenseFuse's forward up to the fused features (`core/model.py:165-186`, `core/fusion.py:21-29` element_fusion 'sum'), round 6: both images of
 * a pair through ONE shared encoder AND `sum` = out_a + out_b (bf16 values added in fp32, one rounding: bit-identical to mmif_fuse_elem_fwd
 * on the two outputs) in one launch -- a wave carries the same 32-column strip of both images and holds both results of a pixel when it
 * stores them, so the separate pass over 192 channel planes disappears.  enc_a / enc_b must name the SAME weights (pointers equal) and
 * differ only in `img`; out_a, out_b as for mmif_dense_encoder_fwd; sum: bf16 halo-0 view of 8 channel blocks.  _supported: 1 when the call
 * is taken (shared weights, the round-5 streaming kernel enabled), else the caller runs mmif_dense_encoder_fwd + mmif_fuse_elem_fwd. */
int32_t mmif_dense_encoder_fwd_sum_supported(const mmif_dense_encoder* enc_a, const mmif_dense_encoder* enc_b, int32_t n, int32_t h, int32_t w);
int mmif_dense_encoder_fwd_sum(const mmif_dense_encoder* enc_a, const mmif_tensor* out_a, const mmif_dense_encoder* enc_b,
                               const mmif_tensor* out_b, const mmif_tensor* sum, void* stream);

/* Weight gradients of the same encoder, all four layers in ONE pass (csrc/enc_wgrad.hip; replaces three mmif_conv2d_reflect_wgrad
 * calls + mmif_conv2d_image_in_wgrad): x = the forward's [x0 | x1 | x2 | ..] (bf16, halo 0, >= 6 channel blocks), gz = the four
 * pre-activation gradients [g0 | g1 | g2 | g3] (bf16, 8 blocks; halo 0, or halo 1 folded).  dw0 [16][1][3][3], dwK [16][16K][3][3]
 * (K = 1..3), dbK [16] (NULL = not wanted).  accumulate != 0 adds to the destinations (shared encoders: second branch).  The first
 * layer runs on the exact fp32 matrix path against the fp32 image.  Deterministic (fixed-order reduction of per-block partials).
 * fp32 tensors: the same call -- the first layer on mmif_conv2d_image_in_wgrad's kernel, the three DenseBlock convs in one split-operand
 * pass (csrc/conv_x3.hip, wgrad_x3_dense_kernel). */
size_t mmif_dense_encoder_wgrad_workspace(void);
int mmif_dense_encoder_wgrad(const float* img, const mmif_tensor* x, const mmif_tensor* gz, float* dw0, float* db0, float* dw1, float* db1,
                             float* dw2, float* db2, float* dw3, float* db3, int32_t accumulate, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Backward chain of the same DenseBlock in GATHER form: the gradient of x_k (k = 2, 1, 0) is ONE dgrad of a virtual layer with 16 input
 * channels (x_k) and 16 (3 - k) output channels -- DenseBlock convs k+1 .. 3 stacked, each restricted to its x_k input slice -- run on
 * the contiguous gradient blocks [g_{k+1} | .. | g_3] with accumulate (onto G_k) + ReLU mask in the epilogue
 * (mmif_conv2d_reflect_dgrad_folded, cin = 16, cout = 16 (3 - k)).  This call writes the three virtual layers' dgrad operand images
 * (sizes mmif_packed_weight_bytes(16 (3 - k), 16, 3)) from the fp32 weights w1 [16][16][3][3], w2 [16][32][3][3], w3 [16][48][3][3]. */
int mmif_pack_dense_chain(const float* w1, const float* w2, const float* w3, void* packed_v0, void* packed_v1, void* packed_v2,
                          void* stream);
/* Both encoder branches of a two-encoder model (PFNetv1: `core/model.py:73-80` twice) in ONE launch: w_x = {w1, w2, w3}, packed_x = {v0, v1, v2}. */
int mmif_pack_dense_chain_pair(const float* const* w_a, void* const* packed_a, const float* const* w_b, void* const* packed_b, void* stream);
/* The whole chain as ONE streaming launch (csrc/enc_chain.hip, bf16): a line-buffer pipeline like mmif_dense_encoder_fwd's, walking down the
 * image -- g2 = [x2 > 0](G2 + A32 g3), g1 = [x1 > 0](G1 + A21 g2 + A31 g3), g0 = [x0 > 0](G0 + A10 g1 + A20 g2 + A30 g3) with the adjoint of
 * reflect padding applied in place (rows 1 / h-2: a second k-loop pass; columns: the edge strips carry columns -1 / w and fold them with one
 * cross-lane add).  g3: 2-block view of the finished gradient of x3; glow: 6-block view G0 | G1 | G2 of the gradient the decoder left
 * (halo 0, or halo 1 folded -- DenseFuse passes the ONE gradient of f1 + f2 for both branches); x: 6-block view x0 | x1 | x2 of the forward's
 * output (halo 0); packed[k]: the dgrad operand image of virtual layer k of mmif_pack_dense_chain; out: 8-block view that receives
 * g0 | g1 | g2 | g3 (halo 0 or 1; it must NOT overlap g3 / glow: strips and row segments recompute their margins from the inputs).  h, w >= 4.
 * chain_b: a second, independent branch in the same launch, or NULL.  Same sums as three mmif_conv2d_reflect_dgrad_folded calls on the
 * virtual layers (fp32 accumulation of every contribution + G, one bf16 rounding); 176 instead of 240 channel planes of HBM traffic. */
typedef struct mmif_dense_chain {
    const mmif_tensor* g3;
    const mmif_tensor* glow;
    const mmif_tensor* x;
    const void* packed[3];
    const mmif_tensor* out;
} mmif_dense_chain;
int mmif_dense_encoder_chain(const mmif_dense_chain* chain_a, const mmif_dense_chain* chain_b, void* stream);
/* Round 5: the WHOLE backward of the DenseBlock encoder -- the gradient chain above AND dW, db of ConvLayer(1, 16) + the three DenseBlock
 * convs (mmif_dense_encoder_wgrad) -- as ONE streaming launch per call (csrc/enc_bwd.hip): g0..g2 never leave the chip, the activations
 * x0..x2 are read once.  chain_x->out is ignored.  dwdb_x = {dW0, db0, dW1, db1, dW2, db2, dW3, db3} (fp32, reference layouts; a db
 * may be NULL); accumulate_x != 0: add onto what they hold (the second branch of a shared encoder).  img_x: the branch's input image
 * [n][h][w] fp32.  workspace: mmif_dense_encoder_bwd_workspace() bytes.  Reference: autograd of core/model.py:73-80 + core/block.py:137-151
 * (train.py:71).  h, w >= 4; bf16 tensors. */
/* 1 when an allocation of cb_total channel blocks of h x w (+ 2 halo) pixels stays within mmif_dense_encoder_bwd's 32-bit lane offsets (one image
 * below 2 GiB) and h, w >= 4: callers ask for every tensor involved and take mmif_dense_encoder_chain + mmif_dense_encoder_wgrad otherwise. */
int32_t mmif_dense_encoder_bwd_fits(int32_t cb_total, int32_t h, int32_t w, int32_t halo);
size_t mmif_dense_encoder_bwd_workspace(void);
int mmif_dense_encoder_bwd(const mmif_dense_chain* chain_a, const float* img_a, float* const* dwdb_a, int32_t accumulate_a,
                           const mmif_dense_chain* chain_b, const float* img_b, float* const* dwdb_b, int32_t accumulate_b,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same for fp32 tensors: x3-format images (mmif_packed_weight_bytes_x3(16 (3 - k), 16, 3)) for the split-operand dgrad kernels. */
int mmif_pack_dense_chain_x3(const float* w1, const float* w2, const float* w3, void* packed_v0, void* packed_v1, void* packed_v2,
                             void* stream);

/* ---- data feed (the step before the hot path).  out[b] = transform(norm(bank[idx[b]]), mode[b]) as fp32 [batch][P][P]:
 *      FusionPatches.__getitem__ data/patches.py:61-74 with norm data/transform.py:15-29 (norm_mode 0: /255.0, 1: 'min-max',
 *      2: 'z-score') and the 8 dihedral variants of transform data/transform.py:38-66 (mode 0..7; NULL = no augmentation), plus
 *      the DataLoader's collate + H2D copy.  bank: uint8 [n_patches][P][P] resident in HBM; idx / mode: int32 [batch], device. */
int mmif_patch_feed(const uint8_t* bank, int64_t n_patches, int32_t patch, const int32_t* idx, const int32_t* mode, int32_t batch,
                    int32_t norm_mode, float* out, void* stream);

/* ---- optimiser (train.py:72-75,319): clip_grad_norm_(max_norm) + Adam on flat fp32 buffers ---- */
size_t mmif_clip_adam_workspace(int64_t numel);
/* grad_scale multiplies the gradients first (1/world after a SUM all-reduce).  max_norm <= 0
 * disables clipping.  norm_out (device float[1], may be NULL) receives the pre-clip global L2 norm. */
int mmif_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel,
                        float lr, float beta1, float beta2, float eps, int32_t step, float max_norm,
                        float grad_scale, float* norm_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- diagnostics: device buffer long long[1024][64]; when non-NULL the MFMA conv kernel stamps s_memtime per
 * phase for its first 1024 blocks (tools/trace_conv.py).  NULL (default) disables it. */
void mmif_debug_set_trace(void* device_buf);
/* kernel-generation switch for cross-checks: 1 (default) = DMA-staged conv / wgrad kernels where they apply,
 * 0 = the register-staged kernels everywhere. */
void mmif_debug_set_conv_dma(int32_t mode);
/* 1 (default) = bf16 1x1 layers (forward, dgrad without an accumulate operand) on the streaming kernel of
 * csrc/conv1x1.hip; 0 = the register-staged conv_mfma_kernel<1, ...>.  Bit-identical results either way. */
void mmif_debug_set_conv1x1_stream(int32_t mode);
/* mmif_conv2d_reflect_bwd_pair: 1 (default) = tiles staged by a loader wave's LDS-DMA into a double-buffered tile
 * (bwd_pair_dma_kernel, round 4), 0 = the register-staged kernel; bit-identical results (tests/test_gpu_bwd_pair.py). */
void mmif_debug_set_bwd_pair_dma(int32_t mode);
/* 3x3 forward with 49..64 input and 17..32 output channels (decode.2 of the PFNet / DenseFuse decoders): 1 (default) = the
 * asynchronous loader / consumer kernel in its two-group, three-slot geometry (round 4), 0 = the register-staged kernel; bit-identical. */
void mmif_debug_set_thin_wide(int32_t mode);
/* persistent blocks (8..256, default 256 = one per CU) of the wide layers' weight-gradient kernel (wgrad_dma_kernel): a smaller grid leaves
 * compute units to a kernel that runs concurrently on another stream -- the intra-step overlap of decode.0's weight gradient with the
 * encoder's backward (round 5's side-stream experiment, measured and closed: DESIGN.md section 4.1).  Results change in the last bits only (the per-block partial sums regroup). */
void mmif_debug_set_wgrad_dma_blocks(int32_t blocks);
/* Weight gradients of 3x3 layers whose channel counts are not multiples of 64 (round 6; NestFuse's 88 / 120 / 136 / 152 / 184 / 304-channel
 * layers): 1 (default, $MMIF_WGRAD_RAGGED) = wgrad_dma_kernel does not stage the channel-block planes of a ragged last group that lie past the
 * tensor (their products are never reduced), 0 = it re-reads the last real plane for them as before.  Identical dW / db either way. */
void mmif_debug_set_ragged(int32_t mode);
/* Deferred weight-gradient reductions (round 5, csrc/reduce_defer.hip).  Every weight-gradient entry point above ends in a small fixed-order
 * reduce launch over its per-block partial sums -- autograd's accumulation of `convolution_backward`'s weight / bias gradients into `.grad`
 * (`train.py:71`).  Between begin() and flush() the reduces of mmif_conv2d_reflect_bwd_wide, mmif_conv2d_reflect_bwd_pair and
 * mmif_conv2d_image_out_wgrad are queued -- their partial sums go to slots of `arena` (device memory, sized by the caller as the sum of the
 * layers' weight-gradient workspaces; a layer that does not fit, or a ninth job, reduces at once as without deferral) -- and flush() runs
 * all of them as ONE launch: the same sums in the same order, bit-identical dW / db, one launch latency instead of five in a PFNetv1 step.
 * dW / db of a queued layer are NOT valid before the flush.  One stream for producers, flush and consumers. */
int mmif_reduce_defer_begin(void* arena, size_t bytes);
int mmif_reduce_defer_flush(int32_t keep_deferring, void* stream);
int32_t mmif_reduce_defer_pending(void);

/* mmif_dense_encoder_fwd: 2 (default) = the round-5 streaming kernel with 32-column strips, eight waves per CU; 1 = the
 * same with 64-column strips, four waves per CU (csrc/enc_stream2.hip: input-stationary accumulation; every stage within one bf16
 * rounding of its fp64 definition); 0 = the round-2 kernel (csrc/enc_stream.hip, bit-identical to the four layer-wise launches). */
void mmif_debug_set_enc_stream2(int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* MMIF_H */
