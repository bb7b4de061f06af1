/*
 * udaseg.h -- C-ABI of libudaseg_hip.so: the MI355X (gfx950) kernels behind the segmentation /
 * domain-adaptation training hot path of bempt/uda_aerial_semantic_segmentation_research.
 *
 * The reference has no FFI layer: its boundary is the torch.nn.Module / callable protocol its trainers
 * use.  Each entry point below therefore names the reference call it replaces (file:line
 * under the reference tree); the Python side (uda_aerial_semantic_segmentation_research_amd/) mirrors the
 * reference's classes on top of these symbols via ctypes.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (PyTorch);
 *    the library allocates nothing and never synchronises the host;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream); all launches are asynchronous;
 *  - activations are NHWC fp32 with a channel count that is a multiple of 4 (images are padded 3->4,
 *    logits 23->24 by udaseg_nchw_to_nhwc / the head conv); conv weights are OHWI [co][kh][kw][ci];
 *  - return 0 on success, a negative UDASEG_E_* otherwise; udaseg_last_error() gives the message
 *    (thread-local).  No C++ exception crosses the boundary.
 */
#ifndef UDASEG_H
#define UDASEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDASEG_OK 0
#define UDASEG_E_BADARG (-1)
#define UDASEG_E_UNSUPPORTED (-2)
#define UDASEG_E_WORKSPACE (-3)
#define UDASEG_E_HIP (-4)

#define UDASEG_ACT_NONE 0
#define UDASEG_ACT_LEAKY 1 /* y = x > 0 ? x : slope * x ; slope 0 => ReLU */

int udaseg_version(void);
/* Runtime switches for cross-checks and tuning: ONE table (csrc/api.hip).  Every key has a default that an environment variable
 * may supply (read once, at first use); udaseg_set_option(key, value) overrides it for the process (value -1: back to the default),
 * udaseg_get_option reads the effective value, udaseg_option_name gives the key's environment variable, udaseg_option_epoch counts
 * the overrides made so far (callers that cache routing answers compare it).  Both gather loops accumulate in the same order:
 * forward and data-gradient results are bit-identical between GENERIC_GATHER 0 and 1. */
#define UDASEG_OPT_GENERIC_GATHER 0        /* UDASEG_IGEMM_GENERIC: 1: the implicit-GEMM kernels keep their generic gather loop (cross-check of the uniform-tap loop); setting this key also sets WGRAD_GENERIC */
#define UDASEG_OPT_F32_SPLIT 1             /* UDASEG_F32_SPLIT: 0: the SHARED-SOURCE fp32 kernels (udaseg_conv2d_fwd / _dgrad / _wgrad) stay on the fp32 matrix pipe */
#define UDASEG_OPT_WGRAD_GENERIC 2         /* UDASEG_WGRAD_GENERIC: 1: the split-K weight gradients keep their generic gather loop */
#define UDASEG_OPT_F32_HALO 3              /* UDASEG_F32_SPLIT: 0: the halo-resident three-term kernels (udaseg_conv2d_*_f32x3, _up_, _n16_) report 'not supported' (A/B: everything on the fp32 pipe) */
#define UDASEG_OPT_F3_CFG 4                /* UDASEG_F3_CFG: conv_halo_f32x3.hip: one tile configuration 1..12 for every launch (udaseg_f32x3_force_config) */
#define UDASEG_OPT_F3_WS 5                 /* UDASEG_F3_WS: 0: the one-role forward kernel everywhere (A/B of the wave-specialised form) */
#define UDASEG_OPT_F3_SIGNS 6              /* UDASEG_F3_SIGNS: 0: no + - - + sign pattern in the split kernels and their packers (bias measurements) */
#define UDASEG_OPT_IGEMM_TILE 7            /* UDASEG_IGEMM_TILE: conv_igemm.hip: 1 (128x128) | 2 (128x64) | 3 (64x64) | 4 (128x32) for every launch */
#define UDASEG_OPT_IGEMM_X3 8              /* UDASEG_IGEMM_X3: three-term mode of the shared implicit GEMM: 0 off | 1 (64x64 tile) | 2 (128x64) | 3 (128x128) */
#define UDASEG_OPT_NO_FOLD 9               /* UDASEG_NO_FOLD: 1: no pixel folding of the <= 32-channel bf16 layers */
#define UDASEG_OPT_WGRAD_X3_BLOCKS 10       /* UDASEG_WGRAD_X3_BLOCKS: blocks of a conv_wgrad_x3_kernel launch */
#define UDASEG_OPT_WGRAD_BLOCKS 11          /* UDASEG_WGRAD_BLOCKS: blocks of a split-K weight-gradient launch (0: 3072 fp32 / the bf16 rules) */
#define UDASEG_OPT_WGRAD_NO_XCD 12          /* UDASEG_WGRAD_NO_XCD: 1: no XCD-aware block order in the bf16 split-K weight gradient */
#define UDASEG_OPT_WGRAD_X3 13              /* UDASEG_WGRAD_X3: 0: conv_wgrad_x3_kernel off (shared-source weight gradients on the fp32 pipe) */
#define UDASEG_OPT_NO_WGRAD_HALO 14         /* UDASEG_NO_WGRAD_HALO: 1: the per-tap split-K weight gradients everywhere */
#define UDASEG_OPT_WGRAD_F3_BLOCKS 15       /* UDASEG_WGRAD_F3_BLOCKS: blocks of an fp32 halo weight-gradient launch */
#define UDASEG_OPT_WGRAD_HALO_BLOCKS 16     /* UDASEG_WGRAD_HALO_BLOCKS: blocks of a bf16 halo weight-gradient launch */
#define UDASEG_OPT_WGRAD_DEEP_BLOCKS 17     /* UDASEG_WGRAD_DEEP_BLOCKS: blocks of the 16-pixel-wide halo weight gradients (0: 128 fp32 / 256 bf16) */
#define UDASEG_OPT_WGRAD_DB 18              /* UDASEG_WGRAD_DB: 0: the single-buffer 4-row form of the 64 x 64 fp32 halo weight gradient */
#define UDASEG_OPT_REDUCE_BLOCKS 19         /* UDASEG_REDUCE_BLOCKS: cap on the blocks of the BatchNorm reduction kernels (0: 512) */
#define UDASEG_OPT_BN_APPLY_PT 20           /* UDASEG_BN_APPLY_PT: vectors per thread of the BatchNorm apply kernels */
#define UDASEG_OPT_GEMM_1X1_TILE 21         /* UDASEG_GEMM_1X1_TILE: conv1x1_gemm_bf16_kernel: 128 | 256 pixel tiles for every launch */
#define UDASEG_OPT_GEMM_1X1 22              /* UDASEG_GEMM_1X1: 0: r50's 1x1 projections never take the small-GEMM kernel */
#define UDASEG_OPT_GEMM_1X1_MAXM 23         /* UDASEG_GEMM_1X1_MAXM: most pixels of a launch the small-GEMM kernel takes */
#define UDASEG_OPT_NO_STREAM 24             /* UDASEG_NO_STREAM: 1: 1x1 layers stay on the tile kernel */
#define UDASEG_OPT_HALO_CFG 25              /* UDASEG_HALO_CFG: conv_halo_bf16.hip: one configuration for every launch */
#define UDASEG_OPT_NO_HALO 26               /* UDASEG_NO_HALO: 1: every bf16 layer on the shared implicit-GEMM source */
#define UDASEG_OPT_NO_HALO_S2 27            /* UDASEG_NO_HALO_S2: 1: the discriminator's 4x4 / stride 2 layers on the shared source */
#define UDASEG_OPT_HALO_W16 28              /* UDASEG_HALO_W16: tile for widths 16 divides and 32 does not: 0 | 4 | 5 | 6 */
#define UDASEG_OPT_HALO_DEEP 29             /* UDASEG_HALO_DEEP: 0: no 8 x 16-pixel tile for deep low-resolution layers */
#define UDASEG_OPT_HALO_S2_CK 30            /* UDASEG_HALO_S2_CK: channels per staged chunk of the 4x4 / stride 2 halo form: 32 | 64 */
#define UDASEG_OPT_UP_CFG 31                /* UDASEG_UP_CFG: conv_up_f32x3.hip: one tile configuration 1..8 for every launch (udaseg_up_f32x3_force_config) */
#define UDASEG_OPT_WGRAD_UP_BLOCKS 32       /* UDASEG_WGRAD_UP_BLOCKS: blocks of a conv_wgrad_up_kernel launch (0: 96; udaseg_wgrad_up_set_blocks) */
#define UDASEG_OPT_COUNT 33
int udaseg_set_option(int key, int value);
int udaseg_get_option(int key);
int udaseg_option_count(void);
const char* udaseg_option_name(int key);
int udaseg_option_epoch(void);
const char* udaseg_last_error(void);
/* number of HIP devices visible to the library (0 on a CPU-only box; never initialises a context) */
int udaseg_device_count(void);
/* Host-side helpers of the launch plan (no reference counterpart: what `torch.zeros` / `stream.wait_stream` do for the reference's
 * eager ops, at a tenth of their host cost -- BASELINE cfg 3 is bound by the host's launch rate).  udaseg_memset_async: byte fill of
 * a device buffer in stream order; udaseg_stream_wait: `waiter` waits for everything enqueued on `signal` so far. */
int udaseg_memset_async(void* ptr, int value, size_t bytes, void* stream);
int udaseg_stream_wait(void* waiter_stream, void* signal_stream);

/* Geometry of one 2-D convolution, NHWC. hi/wi/ci: input; ho/wo/co: output; square stride, symmetric pad.
 * ci and co are the PHYSICAL channel counts (multiples of 4). */
typedef struct {
  int n, hi, wi, ci;
  int ho, wo, co;
  int kh, kw, stride, pad;
} udaseg_conv_desc;

/* ---- convolutions: torch.nn.functional.conv2d reached through smp.Unet.forward (reference
 *      src/models/train.py:341, src/models/adversarial_trainer.py:104) and DomainDiscriminator.forward
 *      (src/models/discriminator.py:54).  Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32). ---- */

/* y[n,ho,wo,co] (+)= conv(x, w) + bias, then optional activation. bias may be NULL. */
int udaseg_conv2d_fwd(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                      int act, float slope, int accumulate, void* stream);
/* y = conv(x, w) + bias AND the training-mode BatchNorm statistics of y in the same pass: stats[R][2][co] f64 (sum, sum of
 * squares; R = udaseg_bn_replicas(), caller-zeroed) -- exactly what udaseg_bn_stats(y) would add, without re-reading y. */
int udaseg_conv2d_fwd_bnstats(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                              double* stats, void* stream);
/* Inference form: y = act(conv(x, w) + bias + residual) in one kernel (bias, residual optional).  With udaseg_bn_fold this
 * is the whole conv+BN(+add)+ReLU block of an eval-mode forward (reference validate(): src/models/train.py:391-438). */
int udaseg_conv2d_fwd_fused(const udaseg_conv_desc* d, const float* x, const float* w, const float* bias,
                            const float* residual, float* y, int act, float slope, void* stream);
/* bf16 storage variants (BASELINE configs 3 and 5): x, w, residual, y are bf16 (channel counts multiples of 8), accumulation,
 * bias and BatchNorm statistics fp32/f64; v_mfma_f32_32x32x16_bf16.  out_f32 != 0 writes y as fp32 (segmentation logits).
 * stats may be NULL. */
int udaseg_conv2d_fwd_bf16(const udaseg_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                           void* y, int out_f32, int act, float slope, double* stats, void* stream);
int udaseg_conv2d_dgrad_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx, int accumulate,
                             void* stream);
/* dw (fp32 master gradient) from bf16 x and dy */
int udaseg_conv2d_wgrad_bf16(const udaseg_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* stream);
/* dx[n,hi,wi,ci] (+)= conv_transpose(dy, w).  w_t is the dgrad packing [ci][kh][kw][co] made by
 * udaseg_pack_dgrad_weights.  Autograd of the convs above: loss.backward() at train.py:343. */
int udaseg_conv2d_dgrad(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx, int accumulate,
                        void* stream);
/* dx = conv_transpose(dy, w) AND, in the same pass, the two reductions of the BatchNorm backward of the layer in front of this
 * convolution (the one whose output prev_y [n][hi][wi][ci] went through training-mode BN + activation to become this
 * convolution's input):  g = dx * act'(gamma*(prev_y-mean)*rstd + beta),  bsums[0][c] += sum g,  bsums[1][c] += sum g*xhat
 * -- exactly what udaseg_bn_bwd_reduce(dz = dx, z = NULL, y = prev_y, ...) adds, without re-reading dx (loss.backward(),
 * reference src/models/train.py:343).  Only for activations with a single consumer (dx is their complete gradient) and for
 * geometries udaseg_conv2d_dgrad_bnreduce_ok() accepts (whole tiles: stride 1, no K-slices, not the small-channel kernel). */
int udaseg_conv2d_dgrad_bnreduce_ok(const udaseg_conv_desc* d);
int udaseg_conv2d_dgrad_bnreduce(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx, const float* prev_y,
                                 const float* save_mean, const float* save_rstd, const float* gamma, const float* beta, int act,
                                 float slope, double* bsums, void* stream);
/* bf16 storage: same reductions from the LDS-staged epilogue (g from the bf16-rounded gradient, as the stand-alone
   udaseg_bn_bwd_reduce_bf16 would read it back); any stride-1 geometry */
int udaseg_conv2d_dgrad_bnreduce_bf16_ok(const udaseg_conv_desc* d);
int udaseg_conv2d_dgrad_bnreduce_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx, const void* prev_y,
                                      const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                                      int act, float slope, double* bsums, void* stream);
/* dw[co][kh][kw][ci] (+)= sum over pixels of dy (x) x.  If !accumulate dw is overwritten.
 * Split-K partials are combined with fp32 atomics. */
int udaseg_conv2d_wgrad(const udaseg_conv_desc* d, const float* x, const float* dy, float* dw, int accumulate,
                        void* stream);
/* ---- fused decoder input: smp's DecoderBlock computes conv(cat([interpolate(a, 2, 'nearest'), skip], 1)) (reference model
 *      created at src/test_system.py:90-95; trace fixture: aten::upsample_nearest2d + aten::cat in front of every decoder
 *      conv1).  The concatenation is never materialised: the convolution reads a [n][hi/2][wi/2][ca] at (iy >> 1, ix >> 1)
 *      for its first ca input channels and skip [n][hi][wi][ci-ca] for the rest (skip == NULL when ca == ci).  d describes
 *      the convolution on the VIRTUAL input (ci = ca + cb, hi x wi = full resolution), stride 1.  Channel counts must be
 *      multiples of the K-tile (32 fp32 / 64 bf16) unless the small-channel kernel applies (ci <= 32, no skip). ---- */
int udaseg_conv2d_fwd_upcat(const udaseg_conv_desc* d, const float* a, const float* skip, int ca, const float* w,
                            const float* bias, float* y, int act, float slope, double* stats, void* stream);
int udaseg_conv2d_fwd_upcat_bf16(const udaseg_conv_desc* d, const void* a, const void* skip, int ca, const void* w,
                                 const float* bias, void* y, int act, float slope, double* stats, void* stream);
/* its data gradient, written straight into the two gradient tensors: dx_a [n][hi][wi][ca] = gradient of the UP-SAMPLED a
 * (the caller reduces it 2x2 with udaseg_upsample2x_concat_bwd, cb = 0), dx_b [n][hi][wi][ci-ca] = gradient of skip.
 * ca must be a multiple of 64. */
int udaseg_conv2d_dgrad_split(const udaseg_conv_desc* d, const float* dy, const float* w_t, float* dx_a, float* dx_b, int ca,
                              void* stream);
int udaseg_conv2d_dgrad_split_bf16(const udaseg_conv_desc* d, const void* dy, const void* w_t, void* dx_a, void* dx_b, int ca,
                                   void* stream);
/* and its weight gradient, one call per source: the columns (tap, c_off + c), c < src_c, of dw[co][kh*kw][d->ci] from the
 * source tensor src ([..][src_c] channels; up != 0: at half resolution behind the nearest x2 up-sampling).  accumulate must
 * be set unless the slice is the whole gradient (src_c == d->ci). */
int udaseg_conv2d_wgrad_part(const udaseg_conv_desc* d, const float* src, int src_c, int c_off, int up, const float* dy,
                             float* dw, int accumulate, void* stream);
int udaseg_conv2d_wgrad_part_bf16(const udaseg_conv_desc* d, const void* src, int src_c, int c_off, int up, const void* dy,
                                  float* dw, int accumulate, void* stream);
/* w[co][kh*kw][ci] -> w_t[ci][kh*kw][co] */
int udaseg_pack_dgrad_weights(const udaseg_conv_desc* d, const float* w, float* w_t, void* stream);
/* The same for every convolution of a network in one launch: table[i] = {src float-offset into arena, dst float-offset
 * into packed, co, kh*kw, ci} (int32, device memory). */
int udaseg_pack_dgrad_batched(const float* arena, float* packed, const int* table, int entries, void* stream);
/* Conv FLOPs (2*MAC) of one call, for roofline accounting. */
double udaseg_conv_flops(const udaseg_conv_desc* d);

/* ---- layout: the DataLoader hands NCHW images (train.py:337); kernels want NHWC with C%4==0 ---- */
/* x[n][c][h][w] -> y[n][h][w][cpad], channels c..cpad-1 zero-filled */
int udaseg_nchw_to_nhwc(const float* x, float* y, int n, int c, int h, int w, int cpad, void* stream);

/* ---- batch norm (training mode) + activation: torch.nn.BatchNorm2d/ReLU/LeakyReLU inside smp.Unet and
 *      discriminator.py:21-33.  sums / bsums = [R][2][c] doubles (R = udaseg_bn_replicas() replicated accumulators of
 *      sum and sum of squares, spread to avoid same-address atomic serialisation), zeroed by the caller. ---- */
int udaseg_bn_replicas(void);
int udaseg_bn_stats(const float* y, int64_t pixels, int c, double* sums, void* stream);
/* the same for a bf16 tensor (channels a multiple of 8): behind the library GEMMs, which have no statistics epilogue */
int udaseg_bn_stats_bf16(const void* y, int64_t pixels, int c, double* sums, void* stream);
/* z = act(gamma*(y-mean)*rstd + beta (+ residual)); writes save_mean/save_rstd [c]; updates running stats
 * (momentum, unbiased variance) when running_mean != NULL.  residual may be NULL. */
int udaseg_bn_apply(const float* y, const double* sums, const float* gamma, const float* beta, const float* residual,
                    float* z, int64_t pixels, int c, float eps, float momentum, float* running_mean,
                    float* running_var, float* save_mean, float* save_rstd, int act, float slope, void* stream);
/* eval mode: z = act(gamma*(y-running_mean)/sqrt(running_var+eps) + beta (+ residual)) */
int udaseg_bn_apply_eval(const float* y, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, const float* residual, float* z, int64_t pixels, int c, float eps,
                         int act, float slope, void* stream);
/* eval mode, folded into the preceding conv: w_folded[co][row_len] = w * gamma/sqrt(var+eps);
 * bias_folded[co] = beta + (bias - mean) * gamma/sqrt(var+eps).  bias may be NULL. */
int udaseg_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, int co, int row_len, float* w_folded, float* bias_folded, void* stream);
/* backward, pass 1: g = dz * act'(z); bsums[0][c] += sum g, bsums[1][c] += sum g*xhat (doubles, caller-zeroed).
 * z may be NULL when the layer had no residual input: the activation's argument is then re-evaluated from y, gamma and
 * beta (the same fused multiply-add as udaseg_bn_apply), which saves one pass over the activation; gamma / beta are only
 * read in that case. */
int udaseg_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* save_mean,
                         const float* save_rstd, const float* gamma, const float* beta, int64_t pixels, int c,
                         double* bsums, int act, float slope, void* stream);
/* backward, pass 2: dy (+)= gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)); dres (+)= g if dres != NULL;
 * dgamma/dbeta (+)= from bsums.  z may be NULL as in pass 1 (then beta is read). */
int udaseg_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* save_mean,
                        const float* save_rstd, const float* gamma, const float* beta, const double* bsums, float* dy,
                        float* dres, float* dgamma, float* dbeta, int64_t pixels, int c, int act, float slope,
                        int accumulate_dy, int accumulate_dres, int accumulate_param, void* stream);
/* plain activation backward for a conv+bias+act epilogue (discriminator layer 1): dy = dz * act'(z) */
int udaseg_act_bwd(const float* dz, const float* z, float* dy, int64_t count, int act, float slope, void* stream);
/* out[c] (+)= sum over pixels of x[p][c]  (bias gradients) */
int udaseg_channel_sum(const float* x, int64_t pixels, int c, float* out,
                       int accumulate, void* stream);
/* same, with the caller's scratch for the partial sums of large inputs (udaseg_channel_sum_scratch_bytes(c) bytes, owned by
   `stream` until the call's work has run); the plain form takes a stream-ordered allocation instead */
int udaseg_channel_sum_ws(const float* x, int64_t pixels, int c, float* out, int accumulate, float* scratch,
                          size_t scratch_bytes, void* stream);
size_t udaseg_channel_sum_scratch_bytes(int c);

/* ---- pooling / resize glue inside smp.Unet ---- */
/* max_pool2d(3, 2, 1): y[n,ho,wo,c], idx = argmax tap (0..8, first max in scan order) */
int udaseg_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, void* stream);
int udaseg_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, int accumulate,
                            void* stream);
/* out[n,2h,2w,ca+cb] = cat(nearest_x2(a[n,h,w,ca]), skip[n,2h,2w,cb]); skip may be NULL (cb = 0) */
int udaseg_upsample2x_concat_fwd(const float* a, const float* skip, float* out, int n, int h, int w, int ca, int cb,
                                 void* stream);
/* da (+)= 2x2 sum of dout[..., :ca]; dskip (+)= dout[..., ca:] */
int udaseg_upsample2x_concat_bwd(const float* dout, float* da, float* dskip, int n, int h, int w, int ca, int cb,
                                 int accumulate_da, int accumulate_dskip, void* stream);

/* The decoder's alternate up-sampling mode (north_star: "bilinear-upsample + conv"; the reference's traced model uses nearest,
 * SURVEY F5): out = cat(interpolate(a, scale_factor=2, mode="bilinear", align_corners=False), skip) and its backward in
 * gather form.  a / skip / out / gradients are fp32 (channels % 4 == 0) or, with bf16 != 0, bf16 (channels % 8 == 0). */
int udaseg_upsample2x_bilinear_concat_fwd(const void* a, const void* skip, void* out, int n, int h, int w, int ca, int cb,
                                          int bf16, void* stream);
int udaseg_upsample2x_bilinear_concat_bwd(const void* dout, void* da, void* dskip, int n, int h, int w, int ca, int cb,
                                          int accumulate_da, int accumulate_dskip, int bf16, void* stream);

/* ---- per-pixel cross entropy: nn.CrossEntropyLoss() at train.py:208,342 (mean, no weights/ignore) ----
 * logits[p][ldc] with `classes` valid channels; target int64[p]; lse[p] saved for backward;
 * partials: scratch of udaseg_ce_partials() doubles; loss: 1 float. */
int udaseg_ce_partials(void);
int udaseg_ce_fwd(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc, float* lse,
                  double* partials, float* loss, void* stream);
/* dlogits[p][ldc] = (softmax - onehot) * (*grad_out) / pixels; pad channels written as 0.
 * Optional (ldc <= 32): colsum[ldc] = per-class sum of dlogits over all pixels (= the head conv's bias gradient, reference
 * src/models/train.py:343), produced in the same pass; colsum_partials = scratch of udaseg_ce_partials()*ldc floats. */
int udaseg_ce_bwd(const float* logits, const int64_t* target, const float* lse, const float* grad_out, int64_t pixels,
                  int classes, int ldc, float* dlogits, float* colsum_partials, float* colsum, void* stream);

/* round 5: both in ONE pass over the logits (ldc <= 32): loss as udaseg_ce_fwd leaves it (bit for bit), dlogits / colsum as
 * udaseg_ce_bwd leaves them for an upstream gradient of exactly 1 (bit for bit) -- what loss.backward() (train.py:343) passes.
 * udaseg_scale_unless_one(x, count, x2, count2, g): x[i] *= *g, x2[j] *= *g unless the DEVICE scalar *g is 1 (then the launch
 * returns at once): the backward of a loss whose gradient was made ahead of time (count % 4 == 0, count2 <= 256). */
int udaseg_ce_fwd_bwd(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc, double* partials, float* loss,
                      float* dlogits, float* colsum_partials, float* colsum, void* stream);
int udaseg_scale_unless_one(float* x, int64_t count, float* x2, int count2, const float* g, void* stream);

/* ---- the reference's other segmentation losses (src/models/losses.py), same logits layout as udaseg_ce_* (ldc <= 32) ----
 * udaseg_seg_partials(): doubles of scratch the *_fwd calls below need in `partials`. */
int udaseg_seg_partials(void);
/* DiceLoss (losses.py:110-152): softmax over classes, per image b and class c
 *   I = sum_pix p_c*[t==c], U = sum_pix p_c + sum_pix [t==c]; loss = 1 - mean_{b,c} (2I+smooth)/(U+smooth).
 * pooled != 0: the segmentation_models_pytorch DiceLoss(mode='multiclass') form used by the reference's UDALoss
 *   (src/models/uda.py:84): I, U summed over the batch too, score_c = (2I+smooth)/max(U+smooth, eps),
 *   loss = mean_c (1-score_c)*[class c occurs in target]   (smp defaults: smooth 0, eps 1e-7).
 * sums: batch*3*classes doubles, caller-zeroed; coef: batch*2*classes floats kept for udaseg_dice_bwd. */
int udaseg_dice_fwd(const float* logits, const int64_t* target, int batch, int64_t pix_per_image, int classes, int ldc,
                    float smooth, float eps, int pooled, double* sums, float* coef, float* loss, void* stream);
/* dlogits (+)= (*grad_out) * weight * dLoss/dlogits (grad_out may be NULL = 1); pad channels written as 0 */
int udaseg_dice_bwd(const float* logits, const int64_t* target, const float* coef, const float* grad_out, float weight,
                    int batch, int64_t pix_per_image, int classes, int ldc, float* dlogits, int accumulate, void* stream);
/* focal term of WeightedSegmentationLoss (losses.py:181-197): ce = w[t]*(-log softmax_t) (class_weights may be NULL),
 * pt = exp(-ce), f = alpha*(1-pt)^gamma*ce; loss (+)= mean (mean=1) or sum (mean=0) of f over pixels */
int udaseg_focal_fwd(const float* logits, const int64_t* target, const float* class_weights, float alpha, float gamma,
                     int64_t pixels, int classes, int ldc, int mean, double* partials, float* loss, int accumulate,
                     void* stream);
/* dlogits (+)= (*grad_out) * weight * df/dlogits per pixel (the caller folds 1/pixels into weight for 'mean') */
int udaseg_focal_bwd(const float* logits, const int64_t* target, const float* class_weights, float alpha, float gamma,
                     const float* grad_out, float weight, int64_t pixels, int classes, int ldc, float* dlogits,
                     int accumulate, void* stream);
/* ConsistencyLoss (losses.py:53-108): p_i = softmax(z_i / T);
 * loss = (KL(p2||p1) + KL(p1||p2)) / (2 * batch)  (F.kl_div(..., reduction='batchmean') both ways, averaged) */
int udaseg_consistency_fwd(const float* z1, const float* z2, float temperature, int batch, int64_t pixels, int classes,
                           int ldc, double* partials, float* loss, void* stream);
/* gradients with respect to BOTH predictions (the reference detaches neither); d1 or d2 may be NULL */
int udaseg_consistency_bwd(const float* z1, const float* z2, float temperature, const float* grad_out, float weight,
                           int batch, int64_t pixels, int classes, int ldc, float* d1, float* d2, int accumulate,
                           void* stream);

/* ---- validation metrics (SegmentationTrainer.calculate_metrics, train.py:225-243; src/analysis/metrics.py:17-29):
 * confusion[t*classes + argmax(logits[p])] += 1 over all pixels (int64, caller-zeroed); pred[p] = argmax (optional).
 * classes <= 32, ldc <= 32. */
int udaseg_argmax_confusion(const float* logits, const int64_t* target, int64_t pixels, int classes, int ldc,
                            int64_t* confusion, int64_t* pred, void* stream);

/* ---- discriminator tail + adversarial BCE: discriminator.py:37-42, losses.py:18-51 ---- */
/* pooled[n][c] = mean over hw of z; p[n] = sigmoid(dot(pooled[n], w) + b).  partial: [n][splits][c] floats */
int udaseg_gap_splits(int hw);
int udaseg_gap_linear_sigmoid_fwd(const float* z, const float* w, const float* b, float* partial, float* pooled,
                                  float* p, int n, int hw, int c, void* stream);
/* given dp[n]: dw (+)=, db (+)=, dz[n][hw][c] = dp*p*(1-p)*w[c]/hw broadcast */
int udaseg_gap_linear_sigmoid_bwd(const float* dp, const float* p, const float* pooled, const float* w, float* dz,
                                  float* dw, float* db, int n, int hw, int c, int accumulate_param, void* stream);
/* feature-level discriminator tail Conv2d(c,1,1) -> AdaptiveAvgPool2d(1) (src/models/uda.py:22-23):
 * logit[n] = dot(mean_hw z[n], w) + b  (no sigmoid; that design applies BCE-with-logits to real logits) */
int udaseg_gap_linear_fwd(const float* z, const float* w, const float* b, float* partial, float* pooled, float* logit, int n,
                          int hw, int c, void* stream);
int udaseg_gap_linear_bwd(const float* dlogit, const float* pooled, const float* w, float* dz, float* dw, float* db, int n,
                          int hw, int c, int accumulate_param, void* stream);
/* loss (+)= weight * mean(softplus(x) - x*label)   (BCEWithLogits on whatever x is: reference feeds probabilities) */
int udaseg_bce_logits_fwd(const float* x, int n, float label, float weight, float* loss, int accumulate, void* stream);
/* dx = (*grad_out) * weight * (sigmoid(x) - label) / n */
int udaseg_bce_logits_bwd(const float* x, int n, float label, float weight, const float* grad_out, float* dx,
                          int accumulate, void* stream);
/* the same with a per-sample target vector (nn.BCEWithLogitsLoss()(x, y): src/models/uda.py:85,96; trainer_phases.py:157) */
int udaseg_bce_logits_target_fwd(const float* x, const float* target, int n, float weight, float* loss, int accumulate,
                                 void* stream);
int udaseg_bce_logits_target_bwd(const float* x, const float* target, int n, float weight, const float* grad_out, float* dx,
                                 int accumulate, void* stream);

/* ---- Adam: torch.optim.Adam(...).step() at train.py:344,461; adversarial_trainer.py:56-59,98,114 ----
 * flat fp32 arrays; bc1 = 1-beta1^t, bc2 = 1-beta2^t computed by the caller. */
int udaseg_adam_flat(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                     float eps, float bc1, float bc2, void* stream);

/* ---- device-side input pipeline: uint8 RGB HWC images [n][h][w][3] (+ uint8 masks [n][h][w], may be NULL) ->
 *      normalised, D4-augmented, channel-padded NHWC model input (fp32, or bf16 when out_bf16) and int64 masks.
 *      Replaces the per-sample host work of src/data/dataset.py:116-138 with the geometric part of
 *      src/models/augmentation.py:11-13 (RandomRotate90 / Flip / Transpose = one D4 element per sample) and
 *      A.Normalize() (augmentation.py:36): out = (x - mean255[c]) * inv_std255[c] in fp32.
 *      d4[n] (may be NULL = identity): bit0 transpose, bit1 flip rows, bit2 flip columns, applied in that order.
 *      mean255 / inv_std255 are HOST arrays of 3 floats.  Transposing codes need h == w. ---- */
int udaseg_prepare_batch_u8(const uint8_t* images, const uint8_t* masks, const int32_t* d4, int n, int h, int w,
                            const float* mean255, const float* inv_std255, void* out_images, int cpad, int out_bf16,
                            int64_t* out_masks, int square_checked, void* stream);

/* ---- scratch: one caller-owned device buffer PER DEVICE the library may use for split partial results (currently the
 *      small-channel weight gradient, <= 10 MiB); the call binds it to the device that is current when it is made.
 *      Without it those calls take the generic atomics path. ---- */
int udaseg_set_workspace(void* ptr, size_t bytes);
/* bytes of that buffer udaseg_conv2d_wgrad would use for this convolution (0 = none); the maximum over a network's layers is
 * what the caller should provide (the Python host hands over 16 MiB once per device). */
size_t udaseg_workspace_bytes(const udaseg_conv_desc* d);

/* ---- bf16 storage path (BASELINE configs 3 / 5): the HBM-bound kernels above on bf16 NHWC tensors (channels % 8 == 0),
 *      arithmetic and statistics in fp32 / f64; same meaning as their fp32 namesakes.  Parameters, their gradients and the
 *      optimizer state stay fp32 (master copies); udaseg_cast_f32_to_bf16 makes the per-step bf16 weight copy. ---- */
int udaseg_bn_apply_bf16(const void* y, const double* sums, const float* gamma, const float* beta, const void* residual,
                         void* z, int64_t pixels, int c, float eps, float momentum, float* running_mean, float* running_var,
                         float* save_mean, float* save_rstd, int act, float slope, void* stream);
int udaseg_bn_bwd_reduce_bf16(const void* dz, const void* z, const void* y, const float* save_mean, const float* save_rstd,
                              int64_t pixels, int c, double* bsums, int act, float slope, void* stream);
int udaseg_bn_bwd_apply_bf16(const void* dz, const void* z, const void* y, const float* save_mean, const float* save_rstd,
                             const float* gamma, const double* bsums, void* dy, void* dres, float* dgamma, float* dbeta,
                             int64_t pixels, int c, int act, float slope, int accumulate_dy, int accumulate_dres,
                             int accumulate_param, void* stream);
int udaseg_act_bwd_bf16(const void* dz, const void* z, void* dy, int64_t count, int act, float slope, void* stream);
int udaseg_channel_sum_bf16(const void* x, int64_t pixels, int c, float* out, int accumulate, void* stream);
int udaseg_channel_sum_bf16_ws(const void* x, int64_t pixels, int c, float* out, int accumulate, float* scratch,
                               size_t scratch_bytes, void* stream);
int udaseg_nchw_to_nhwc_bf16(const float* x, void* y, int n, int c, int h, int w, int cpad, void* stream);
int udaseg_cast_f32_to_bf16(const float* x, void* y, int64_t count, void* stream);
int udaseg_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, void* stream);
int udaseg_maxpool3x3s2_bwd_bf16(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c, int accumulate,
                                 void* stream);
int udaseg_upsample2x_concat_bwd_bf16(const void* dout, void* da, void* dskip, int n, int h, int w, int ca, int cb,
                                      int accumulate_da, int accumulate_dskip, void* stream);
int udaseg_gap_partial_bf16(const void* z, float* partial, int n, int hw, int c, void* stream);
int udaseg_gap_finish(const float* partial, const float* w, const float* b, float* pooled, float* p, int n, int hw, int c,
                      void* stream);
int udaseg_gap_bwd_broadcast_bf16(const float* dp, const float* p, const float* w, void* dz, int n, int hw, int c, void* stream);
int udaseg_gap_bwd_param(const float* dp, const float* p, const float* pooled, float* dw, float* db, int n, int c,
                         int accumulate_param, void* stream);
int udaseg_pack_dgrad_batched_bf16(const float* arena, void* packed, const int* table, int entries, void* stream);

/* ---- bf16-first convolution kernels (round 3; csrc/conv_halo_bf16.hip): stride-1 3x3 / pad 1 and 1x1 / pad 0 convolutions of
 *      smp.Unet.forward and their data gradients (reference src/models/train.py:341,343; src/models/adversarial_trainer.py:104,113)
 *      for BASELINE configs 3 and 5.  The block stages the input halo of a channel chunk once into LDS, the weights arrive
 *      pre-packed in MFMA-fragment order and never touch LDS.
 *
 *      Fragment packing of a convolution with N produced / K gathered channels and a ks x ks window:
 *        packed[nb][dx][k16][dy][lane][j]  (nb < ceil(N/32), k16 < ceil(K/16), lane < 64, j < 8; bf16; udaseg_frag_elems elements)
 *          = Wsrc[n = 32 nb + (lane & 31)][tap][k = 16 k16 + 8 (lane >> 5) + j]   (zero where n >= N or k >= K)
 *      forward: Wsrc = the bf16 OHWI weights (N = co, K = ci, tap = dy*ks + dx); data gradient: Wsrc = the bf16 dgrad packing
 *      [ci][taps][co] (N = ci, K = co) with the window flipped.  udaseg_pack_frag_batched_bf16 packs every convolution of a
 *      network in one launch: table[i] = {mode (0 forward from w16 / 1 data gradient from wt16), source element offset,
 *      destination element offset, N, K, ks} (int32 x 6, device memory). ---- */
int64_t udaseg_frag_elems(int n_out, int k_in, int ks);
/* STREAM CONTRACT of the two per-device scratch bindings (udaseg_set_workspace, udaseg_set_stats_scratch): they are keyed by
 * DEVICE, not by stream.  Launches that use them -- K-sliced / split partial results, and the statistics of launches with more
 * than 1024 blocks -- must all be issued on ONE compute stream per device (or be ordered by the caller): two such launches on
 * different streams of one device would add into the same scratch and the first fold would take the second's partials.  The
 * weight-gradient side stream of the Python host (engine.py) only issues launches that use neither.  Round 5: for the STATISTICS
 * scratch the contract is enforced -- the first stream that needs it after a bind owns it; a launch on any other stream of that
 * device does not get it and adds into the 16 replicas of `stats` directly (correct, slower).
 * Optional caller-owned f64 scratch (bound to the current device; the caller ZEROES it once, the library leaves it zeroed after
 * every use): launches of more than 1024 blocks put their per-channel statistics there (256 replicas) and a fold kernel adds
 * them into `stats` -- otherwise hundreds of same-address f64 atomics per accumulator bound the low-channel full-resolution
 * layers.  Needs 256 * 2 * co * 8 bytes for a layer with co output channels (256 KiB covers co <= 64); NULL unbinds. */
int udaseg_set_stats_scratch(void* ptr, size_t bytes);
int udaseg_pack_frag_batched_bf16(const void* w16, const void* wt16, void* packed, const int* table, int entries, void* stream);
/* ---- a BatchNorm + activation that is never written (single-consumer layers, bf16): the producer's statistics are finalised
 *      into per-channel scale = gamma * rstd and shift = beta - mean * scale (plus the saved mean / rstd and the running
 *      statistics, exactly as udaseg_bn_apply_bf16 would), and every consumer of the activation applies
 *      act(fma(y, scale, shift)) rounded to bf16 while it stages y: the forward convolution (udaseg_conv2d_fwd_frag_bf16
 *      in_scale / in_shift), its weight gradient (udaseg_conv2d_wgrad_bnin_bf16) and the layer's own BatchNorm backward
 *      (udaseg_bn_bwd_apply_recompute_bf16: the mask is re-evaluated with the same fused multiply-add; its two reductions come
 *      from the consumer's data gradient, udaseg_conv2d_dgrad_frag_bf16 prev_y).  nn.BatchNorm2d + nn.ReLU inside smp.Unet's
 *      blocks, reference src/models/train.py:341,343. ---- */
int udaseg_bn_finalize(const double* sums, const float* gamma, const float* beta, int64_t pixels, int c, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_rstd, float* scale, float* shift,
                       void* stream);
int udaseg_bn_bwd_apply_recompute_bf16(const void* dz, const void* y, const float* fwd_scale, const float* fwd_shift,
                                       const float* save_mean, const float* save_rstd, const float* gamma, const double* bsums,
                                       void* dy, float* dgamma, float* dbeta, int64_t pixels, int c, int act, float slope,
                                       void* stream);
/* Weight gradient of a stride-1 3x3 / pad 1 convolution, bf16 operands, fp32 dW ACCUMULATED onto (caller-zeroed arena): a block
 * owns a 64 x 64 (or 32 x 64, 32 x 32) channel block of all nine taps and walks pixel tiles, x halo and dy tile staged once per
 * tile; few long-lived blocks, one set of atomics per wave (round 4: csrc/conv_wgrad_halo2.hip, shared with the fp32 split form
 * below; channel counts multiples of 32, images >= 32 pixels wide or 16-pixel-wide with 64-multiples).  skip / up_ca: the two
 * sources of a fused decoder input in ONE launch.  loss.backward() at reference src/models/train.py:343. */
int udaseg_conv2d_wgrad_halo_bf16_ok(const udaseg_conv_desc* d, int up_ca);
int udaseg_conv2d_wgrad_halo_bf16(const udaseg_conv_desc* d, const void* x, const void* skip, int up_ca, const void* dy, float* dw,
                                  void* stream);
int udaseg_conv2d_wgrad_bnin_bf16(const udaseg_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, int in_act,
                                  float in_slope, const void* dy, float* dw, int accumulate, void* stream);
/* 1 when the convolution (dgrad = 0: forward, gathers ci and produces co; dgrad = 1: its data gradient; up_ca > 0: forward on the
 * fused decoder input cat([nearest_x2(a), skip]) with up_ca channels from a) can take these kernels */
int udaseg_conv_frag_ok(const udaseg_conv_desc* d, int dgrad, int up_ca);
/* 1 when, in addition, they are expected to be FASTER than the shared implicit-GEMM source for this shape (measured heuristic:
 * csrc/conv_halo_bf16.hip halo_choice) -- what the Python side asks before it routes a layer */
int udaseg_conv_frag_preferred(const udaseg_conv_desc* d, int dgrad, int up_ca);
/* y = act(conv(X, w) + bias) (+ BatchNorm statistics of conv(X, w) + bias into stats, as udaseg_conv2d_fwd_bnstats).
 * X = x [n][h][w][ci], or with up_ca > 0 the virtual cat([nearest_x2(x [n][h/2][w/2][up_ca]), skip [n][h][w][ci - up_ca]]), or with
 * in_scale / in_shift (fp32 [ci]) the producer's BatchNorm + activation applied on the fly: X = in_act(x * in_scale + in_shift)
 * rounded to bf16 -- the normalised activation of a single-consumer layer never reaches HBM.  out_f32: y is fp32 (logits).
 * Round 4: (1) 1x1 / stride-1 launches that are small GEMMs (ci, co multiples of 64 and >= 128, at most 73728 pixels: r50's
 * bottleneck projections from 96^2 down, reference smp.Unet("resnet50"), src/test_system.py:90-95) run on conv1x1_gemm_bf16_kernel
 * (LDS-DMA ring, persistent blocks), same epilogue options; (2) the descriptor may be the discriminator's 4x4 / stride 2 / pad 1
 * convolution (reference src/models/discriminator.py:15-34): x is [n][hi][wi][ci] with ci a power of two, y [n][hi/2][wi/2][co],
 * wfrag the 2x2-window packing over the input's four parity phases (table row mode 2: udaseg_frag_elems(co, 4 * ci, 2) elements);
 * no skip / in_scale there.  The data gradient takes the same descriptor with wfrag_t = the four parity-class packings (modes
 * 3, 4, 5, 6 = input row / column parity (0,0), (0,1), (1,0), (1,1)), udaseg_frag_elems(ci, co, 2) elements each, one after the other. */
int udaseg_conv2d_fwd_frag_bf16(const udaseg_conv_desc* d, const void* x, const void* skip, int up_ca, const void* wfrag,
                                const float* bias, const float* in_scale, const float* in_shift, int in_act, float in_slope,
                                void* y, int out_f32, int act, float slope, double* stats, void* stream);
/* dx = conv_transpose(dy, w) from the data-gradient fragment packing.  split > 0: channels [0, split) of the gradient go to dx
 * [n][h][w][split], the rest to dx2 (the two sources of a fused decoder input).  prev_y != NULL: also the BatchNorm-backward
 * reductions of the layer behind (as udaseg_conv2d_dgrad_bnreduce_bf16).  accumulate: dx += (one rounding of the fp32 sum). */
int udaseg_conv2d_dgrad_frag_bf16(const udaseg_conv_desc* d, const void* dy, const void* wfrag_t, void* dx, void* dx2, int split,
                                  const void* prev_y, const float* save_mean, const float* save_rstd, const float* gamma,
                                  const float* beta, int bn_act, float bn_slope, double* bsums, int accumulate, void* stream);

/* ---- fp32 convolutions on the bf16 matrix pipe (csrc/conv_halo_f32x3.hip): fp32 NHWC tensors in and out, every operand split
 *      exactly into three bf16 terms (x = bf16(x) + bf16(x - x0) + bf16(x - x0 - x1)), the six products with i + j <= 2 on
 *      v_mfma_f32_32x32x16_bf16, fp32 accumulation -- one unit in the last place of each PRODUCT is left out, below the
 *      rounding of the fp32 accumulation itself.  Stride-1 3x3 / pad 1 layers of smp.Unet (reference src/models/train.py:341,
 *      343) in the fp32 configurations; replaces udaseg_conv2d_fwd_bnstats / _upcat / udaseg_conv2d_dgrad(_bnreduce / _split)
 *      there.  UDASEG_F32_SPLIT=0 keeps every layer on the fp32-MFMA kernels.
 *      Weights: three planes of the fragment packing above (plane stride udaseg_frag_elems(N, K, 3) bf16 elements), made by
 *      udaseg_pack_frag_batched_f32x3 from the fp32 OHWI arena (mode 0) / the fp32 dgrad packing [ci][taps][co] (mode 1);
 *      table rows as for udaseg_pack_frag_batched_bf16, dst offset = plane 0 of the entry's 3 * frag_elems block. ---- */
int udaseg_pack_frag_batched_f32x3(const float* w32, const float* wt32, void* packed, const int* table, int entries, void* stream);
/* can / should this (forward or data-gradient) launch take the split kernel?  preferred = the library's measured heuristic */
int udaseg_conv_f32x3_ok(const udaseg_conv_desc* d, int dgrad, int up_ca);
int udaseg_conv_f32x3_preferred(const udaseg_conv_desc* d, int dgrad, int up_ca);
/* tests / tuning: 1 = 8 x 32 pixels x 32 channels per block, 2 = x 64 channels, for every launch; 0 = the heuristic again */
int udaseg_f32x3_force_config(int cfg);
/* y = act(conv(x) + bias); up_ca > 0: x is the half-resolution tensor of a fused decoder input cat([nearest_x2(x), skip]);
 * stats != NULL: BatchNorm statistics of y ([R][2][co] f64, accumulated) */
int udaseg_conv2d_fwd_f32x3(const udaseg_conv_desc* d, const float* x, const float* skip, int up_ca, const void* wfrag3,
                            const float* bias, float* y, int act, float slope, double* stats, void* stream);
/* The same BatchNorm + activation that is never written, on fp32 tensors (round 4; the <= 32-channel full-resolution decoder layers,
 * where the stand-alone udaseg_bn_apply pass moves 268 MB per layer at 8 x 512^2): x is the producer's RAW convolution output, the
 * staging applies act(fma(x, in_scale[c], in_shift[c])) -- udaseg_bn_apply's own arithmetic, bit for bit -- before the split, zero
 * padding stays zero.  up != 0: x is the half-resolution tensor [n][hi/2][wi/2][ci] behind a nearest x2 up-sampling (a decoder
 * block without a skip input).  Forward: every geometry of udaseg_conv2d_fwd_f32x3 with a single source.  Weight gradient: the
 * small-channel direct kernel (<= 32 channels, fp32 pipe; also up) or the halo-resident split kernel (plain source); dW accumulated
 * or overwritten.  The producer's own BatchNorm backward needs nothing new: on fp32 it re-evaluates the activation's
 * argument from its conv output already (udaseg_bn_bwd_reduce / _apply with z = NULL).  nn.BatchNorm2d + nn.ReLU in front of a
 * 3x3 convolution inside smp.Unet's decoder blocks and head, reference src/models/train.py:341,343. */
int udaseg_conv2d_fwd_f32x3_bnin_ok(const udaseg_conv_desc* d, int up);
/* z_out != NULL (plain source, launches for which udaseg_conv2d_fwd_f32x3_bnin_writes says 1: the wave-specialised kernel): the
 * loader waves also WRITE the transformed activation ([n][hi][wi][ci], bit for bit udaseg_bn_apply's output) -- the convolution's
 * weight gradient then reads it like any activation; saved: bn_apply's launch and its read of x */
int udaseg_conv2d_fwd_f32x3_bnin_writes(const udaseg_conv_desc* d);
int udaseg_conv2d_fwd_f32x3_bnin(const udaseg_conv_desc* d, const float* x, int up, const float* in_scale, const float* in_shift,
                                 int in_act, float in_slope, float* z_out, const void* wfrag3, const float* bias, float* y, int act,
                                 float slope, double* stats, void* stream);
int udaseg_conv2d_wgrad_bnin_ok(const udaseg_conv_desc* d, int up);
int udaseg_conv2d_wgrad_bnin(const udaseg_conv_desc* d, const float* x, int up, const float* in_scale, const float* in_shift,
                             int in_act, float in_slope, const float* dy, float* dw, int accumulate, void* stream);
/* dx (+)= conv_transpose(dy, w); split / prev_y / accumulate as udaseg_conv2d_dgrad_frag_bf16, on fp32 tensors */
int udaseg_conv2d_dgrad_f32x3(const udaseg_conv_desc* d, const float* dy, const void* wfrag3_t, float* dx, float* dx2, int split,
                              const float* prev_y, const float* save_mean, const float* save_rstd, const float* gamma,
                              const float* beta, int bn_act, float bn_slope, double* bsums, int accumulate, void* stream);

/* dW[co][9][ci] += weight gradient of a stride-1 3x3 layer, fp32 x / dy with the same three-term split (round 4:
 * csrc/conv_wgrad_halo2.hip, conv_wgrad_h2_kernel -- conflict-free 32-channel LDS sub-planes, one x fragment shared by the three
 * taps of a kernel column, double-buffered 2-row tiles).  Channel
 * blocks: 64 x 64, 32 produced x 64 gathered, 32 x 32 (co % 32 == 0 and ci % 32 == 0); images at least 32 pixels wide, or 16-pixel-
 * wide ones with 64-multiples (8 x 16 tiles).  Reference: loss.backward(), src/models/train.py:343.
 * up_ca > 0: x is the half-resolution tensor of a fused decoder input, skip the other source (up_ca a multiple of the gathered
 * channel block). */
int udaseg_conv2d_wgrad_halo_f32x3_ok(const udaseg_conv_desc* d, int up_ca);
int udaseg_conv2d_wgrad_halo_f32x3(const udaseg_conv_desc* d, const float* x, const float* skip, int up_ca, const float* dy,
                                   float* dw, void* stream);

/* ---- the decoder's up-sampled input convolved as what it is (round 5, csrc/conv_up_f32x3.hip).  smp's DecoderBlock runs
 *      conv3x3(cat([nearest_x2(a), skip])) (reference: model created src/test_system.py:90-95, called src/models/train.py:341,
 *      differentiated :343; trace fixture tests/golden/unet_r50_trace.json: aten::upsample_nearest2d + aten::cat in front of every
 *      decoder conv1).  On the up-sampled channels the nine taps of an output pixel of parity phase (oy & 1, ox & 1) read FOUR
 *      distinct pixels of `a`: a 2 x 2 convolution of `a` per phase with weights pre-summed over the taps that coincide --
 *      4/9 of the multiplications, forward and data gradient, and the data gradient comes out at a's own resolution (no 2 x 2
 *      sum-pool pass).  fp32 tensors, the three-term split of udaseg_conv2d_fwd_f32x3; the pre-sums are fp32 additions in the
 *      packer.  The skip half of the concatenation is a plain 3x3 convolution of `skip` (udaseg_conv2d_fwd_f32x3 /
 *      udaseg_conv2d_dgrad_f32x3 on weight slices packed by modes 0 / 1 below), run FIRST; the up half accumulates on top.
 *      d: the whole convolution at the OUTPUT resolution (hi x wi = 2h x 2w, ci = up_ca + skip channels, co).
 *      udaseg_pack_up_batched_f32x3: table rows of 8 int32 {mode, src element offset, dst element offset (plane 0), N, K, ldk, 0, 0}
 *        mode 0: 3x3 forward packing of channels [offset, offset + K) of OHWI rows of ldk channels (w32), N = co;
 *        mode 1: 3x3 data-gradient packing from wt32 [N][9][K] (ldk = K; rows [up_ca, ci) of the dgrad packing);
 *        mode 2: phase packing for udaseg_conv2d_fwd_up_f32x3 from w32 OHWI [N = co][3][3][ldk], channels [0, K = up_ca);
 *        mode 3: phase packing for udaseg_conv2d_dgrad_up_f32x3 from wt32 [ci][9][K = co], rows [0, N = up_ca);
 *        plane stride udaseg_frag_elems(N, K, 3) (modes 0, 1) / udaseg_frag_elems(N, K, 4) (modes 2, 3) bf16 elements. ---- */
int udaseg_conv_up_f32x3_ok(const udaseg_conv_desc* d, int up_ca);
int udaseg_pack_up_batched_f32x3(const float* w32, const float* wt32, void* packed, const int* table, int entries, void* stream);
/* y[n][hi][wi][co] (+)= conv3x3(nearest_x2(a)) over a's up_ca channels (accumulate != 0: on top of the skip half's result);
 * stats != NULL: BatchNorm statistics of y AFTER the accumulation ([R][2][co] f64, accumulated) */
int udaseg_conv2d_fwd_up_f32x3(const udaseg_conv_desc* d, const float* a, int up_ca, const void* wfrag_up, float* y, int accumulate,
                               double* stats, void* stream);
/* da[n][hi/2][wi/2][up_ca] (+)= gradient of `a` through conv3x3(nearest_x2(a)) given dy[n][hi][wi][co] (co a multiple of 8).
 * prev_y != NULL (the conv output [n][hi/2][wi/2][up_ca] of the conv + BatchNorm + activation layer that produced a, when this
 * convolution is a's only consumer): also that layer's two BatchNorm-backward sums, as udaseg_conv2d_dgrad_f32x3 */
int udaseg_conv2d_dgrad_up_f32x3(const udaseg_conv_desc* d, const float* dy, int up_ca, const void* wfrag_up_t, float* da,
                                 const float* prev_y, const float* save_mean, const float* save_rstd, const float* gamma,
                                 const float* beta, int bn_act, float bn_slope, double* bsums, int accumulate, void* stream);
/* tests / tuning: one tile configuration (1..8, csrc/conv_up_f32x3.hip up_choice) for every launch; 0 = the heuristic again */
int udaseg_up_f32x3_force_config(int cfg);
/* The weight gradient in the same form (csrc/conv_wgrad_halo2.hip, conv_wgrad_up_kernel): the 16 phase-tap correlations
 * T_{py,px}[u][v] = sum_{q,r} dy[2q+py, 2r+px] (x) a[q+py-1+u, r+px-1+v] at a's resolution, each added into the 1, 2 or 4 taps of
 * dW[co][9][ci] it stands for -- 16 tap evaluations per pixel of `a` instead of 36.  dW rows are d->ci channels long, the launch
 * fills channels [0, up_ca) (accumulated onto, like every weight gradient here).  up_ca a multiple of 64, co of 32, a at least 16
 * pixels wide.  The skip half: udaseg_conv2d_wgrad_halo_slice_f32x3 = udaseg_conv2d_wgrad_halo_f32x3 on a channel slice
 * [c_off, c_off + d->ci) of rows of ldw channels (d: the slice as a convolution of its own).  loss.backward(), src/models/train.py:343. */
int udaseg_conv2d_wgrad_up_f32x3_ok(const udaseg_conv_desc* d, int up_ca);
int udaseg_conv2d_wgrad_up_f32x3(const udaseg_conv_desc* d, const float* a, int up_ca, const float* dy, float* dw, void* stream);
int udaseg_conv2d_wgrad_halo_slice_f32x3(const udaseg_conv_desc* d, const float* x, const float* dy, float* dw, int ldw, int c_off,
                                         void* stream);
/* tests / tuning: blocks per launch of the phase-form weight gradient (0 = the default, 128) */
int udaseg_wgrad_up_set_blocks(int blocks);

/* ---- sixteen produced channels on a sixteen-wide matrix tile (round 5, csrc/conv_n16_f32x3.hip): the full-resolution tail of
 *      smp.Unet's decoder (decoder_channels[-1] = 16: block 4 conv2 forward and data gradient, the head's data gradient; reference
 *      src/test_system.py:90-95, src/models/train.py:341,343).  v_mfma_f32_16x16x32_bf16 with K = two taps of a 16-channel chunk:
 *      0.28 of the matrix-pipe cycles of the 32-row tile these layers half-filled.  fp32 tensors, three-term split; stride-1 3x3,
 *      produced channels == 16, gathered channels a multiple of 8 up to 32.  Weight fragments: udaseg_pack_up_batched_f32x3 mode 4
 *      (forward, from the OHWI arena) / mode 5 (data gradient, from the dgrad packing), 3 planes of ceil(K / 16) * 5 * 512 bf16. ---- */
int udaseg_conv_n16_f32x3_ok(const udaseg_conv_desc* d, int dgrad);
int udaseg_conv2d_fwd_n16_f32x3(const udaseg_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, int in_act,
                                float in_slope, const void* wfrag, float* y, double* stats, void* stream);
int udaseg_conv2d_dgrad_n16_f32x3(const udaseg_conv_desc* d, const float* dy, const void* wfrag_t, float* dx, const float* prev_y,
                                  const float* save_mean, const float* save_rstd, const float* gamma, const float* beta, int bn_act,
                                  float bn_slope, double* bsums, void* stream);

/* ---- the encoder's stem (round 5, csrc/conv_stem_f32x3.hip): 7x7 / stride 2 / pad 3, 4 (3 + padding) -> 64 channels, forward --
 *      torchvision ResNet.conv1 inside smp.Unet (reference src/test_system.py:90-95, src/models/train.py:341).  For one kernel row the
 *      7 taps x 4 channels of an output pixel are 28 contiguous floats of an input row: the kernel stages a band of input rows once
 *      and runs K as 7 rows x 32, no im2col gather.  fp32 tensors, three-term split.  Weights: udaseg_pack_up_batched_f32x3 mode 8
 *      (3 planes of 28 * 512 bf16).  Replaces udaseg_conv2d_fwd_bnstats (conv_igemm_kernel, generic gather) on that layer. ---- */
int udaseg_conv_stem_f32x3_ok(const udaseg_conv_desc* d);
int udaseg_conv2d_fwd_stem_f32x3(const udaseg_conv_desc* d, const float* x, const void* wfrag, float* y, double* stats, void* stream);

/* ---- diagnosis: while a device buffer of 6 * blocks u64 is registered, every implicit-GEMM launch of at most `blocks` blocks
 *      writes per block {entry, first tile load, end of K loop, exit} (100 MHz wall-clock ticks), HW_ID and XCC_ID into it
 *      (tools/igemm_timeline.py).  NULL switches it off.  Not for timed runs. ---- */
int udaseg_debug_set_timeline(void* buffer, int blocks);

/* ---- small utilities ---- */
int udaseg_fill_f32(float* p, int64_t count, float value, void* stream);
int udaseg_axpy_f32(float* y, const float* x, int64_t count, float alpha, void* stream); /* y += alpha*x */
/* p[i] += value, int64: num_batches_tracked of every nn.BatchNorm2d of a network (views of one arena) in one launch; reference
 * src/models/train.py:341 (training-mode forward) */
int udaseg_add_i64(int64_t* p, int64_t count, int64_t value, void* stream);
int udaseg_scale_f32(const float* x, float* y, int64_t count, float alpha, void* stream); /* y = alpha*x: gradient reversal,
                                                                                         * src/models/uda.py:99-111 */

/* ---- live kernel timing for bench.py's roofline leg: HIP events bracket every launch of the conv
 *      kernel families on the launch stream while enabled. ---- */
int udaseg_prof_enable(int on);
int udaseg_prof_reset(void);
/* family: 0 = igemm fwd/dgrad, 1 = wgrad.  Synchronises the recorded events (call outside timed regions). */
int udaseg_prof_read(int family, double* total_ms, double* total_flops, int64_t* launches);
/* Per KERNEL SYMBOL (one id per template instantiation; names match rocprofv3's kernel trace): total HIP-event time,
 * algorithmic GEMM FLOPs (2*M*N*K of every launch) and launch count while profiling was enabled. */
int udaseg_prof_kernel_count(void);
const char* udaseg_prof_kernel_name(int kid);
int udaseg_prof_kernel_read(int kid, double* total_ms, double* total_flops, int64_t* launches);
/* Per-launch records: ms, flops, kind (0 fwd, 1 dgrad, 2 wgrad) and the 11 ints of the conv desc. Returns the count. */
int udaseg_prof_records(int family, int max_records, double* ms, double* flops, int* kind, int* desc11);

#ifdef __cplusplus
}
#endif
#endif /* UDASEG_H */
