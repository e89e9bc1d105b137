/*
 * libjirf_hip.so — C ABI of the MI355X (gfx950) inference forward of the JIRF
 * restoration networks (HINT, ASTv2, AST, FPro).
 *
 * The reference has no FFI: the seam this library replaces is the Python class
 * contract of `network_g` (reference basicsr/models/archs/__init__.py:44-47
 * `define_network`, called from basicsr/models/image_restoration_model.py:141
 * and from every <Task>/test_*.py, e.g. Deraining/test_rain100L.py:117-155):
 * construct from the YAML kwargs, load a dict of arrays, call on an NCHW fp32
 * tensor.  Each entry point below names the reference lines it stands in for.
 *
 * Conventions: plain pointers and sizes only.  Every function returns 0 on
 * success or a negative JIRF_E* code; the message of the last failure on the
 * calling thread is returned by jirf_last_error().  Nothing throws across the
 * ABI.  A jirf_net is bound to one GPU and is not thread-safe: use one net per
 * GPU per host thread/process.  Device pointers are fp32, NCHW, contiguous.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef JIRF_HIP_H
#define JIRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JIRF_ABI_VERSION 1

#define JIRF_OK 0
#define JIRF_EINVAL (-1)     /* bad argument / shape the network cannot take */
#define JIRF_ENOTFOUND (-2)  /* unknown arch or parameter key */
#define JIRF_ESTATE (-3)     /* call order violated (e.g. forward before finalize) */
#define JIRF_EHIP (-4)       /* a HIP runtime call failed */
#define JIRF_ENOMEM (-5)     /* device allocation failed */
#define JIRF_ENOGPU (-6)     /* no usable gfx950 device */

typedef struct jirf_net jirf_net;

/* dtype tags of jirf_net_set_param */
#define JIRF_F32 0
#define JIRF_I32 1
#define JIRF_I64 2
#define JIRF_F64 3

/* batch semantics of jirf_net_forward for B > 1 */
#define JIRF_BATCH_PER_IMAGE 0 /* equals B separate B=1 reference calls (what every test script does) */
#define JIRF_BATCH_COUPLED 1   /* the reference's literal B>1 call: HINT's regroup statistics and
                                  cache use batch means (hint_arch.py:245,282) */

int jirf_abi_version(void);
int jirf_device_count(void);
const char* jirf_last_error(void);

/* ---- network objects ------------------------------------------------------------------- */

/* Stands in for `cls(**opt)` of define_network (archs/__init__.py:22-47) with
 * arch in {"HINT" (ctor kwargs of hint_arch.py:495-508), "ASTv2"
 * (astv2_arch.py:303-315), "AST" (ast_arch.py:941-965; token_mlp "frfn", the
 * only feed-forward its configs use), "FPro" (fpro_arch.py:549-561)}.  cfg_json is the `network_g` YAML block
 * without `type`, as JSON.  Unknown arch: NULL + JIRF_ENOTFOUND; unknown
 * option: NULL + JIRF_EINVAL. */
jirf_net* jirf_net_create(const char* arch, const char* cfg_json, int device);
void jirf_net_destroy(jirf_net* net);

/* Stands in for one entry of `net.load_state_dict(checkpoint["params"])`
 * (Deraining/test_rain100L.py:119-124; loader semantics basicsr/models/base_model.py:228-258).
 * `key` is the reference's state-dict key; `host_data` is a host array of
 * `dtype` with `shape[ndim]`, copied (the caller keeps ownership).  Keys the
 * network does not own return JIRF_ENOTFOUND; a shape mismatch JIRF_EINVAL.
 * `qv_cache` is accepted and ignored, as every test script deletes it. */
int jirf_net_set_param(jirf_net* net, const char* key, const void* host_data, int dtype,
                       const int64_t* shape, int ndim);

/* Number of parameter keys the network expects and the i-th key/shape (for
 * state_dict() and strict-load checks).  shape_out must hold 8 dims. */
int jirf_net_param_count(const jirf_net* net);
int jirf_net_param_info(const jirf_net* net, int index, const char** key_out, int64_t* shape_out,
                        int* ndim_out);
/* Copy a parameter back to the host in its reference layout (fp32). */
int jirf_net_get_param(const jirf_net* net, const char* key, float* host_out, size_t n_floats);

/* Checks that every parameter was set, packs the weights into the kernels'
 * layouts and uploads them.  Stands in for `.cuda(); .eval()` (test_rain100L.py:126-127). */
int jirf_net_finalize(jirf_net* net);

/* Sizes the activation arena for inputs of [B,3,H,W] (allocates device memory
 * on first use or growth; forward never allocates afterwards).  Returns the
 * arena size. */
int jirf_net_plan(jirf_net* net, int B, int H, int W, int batch_mode, size_t* workspace_bytes);

/* Stands in for `restored = model_restoration(input_)` (test_rain100L.py:155;
 * HINT.execute hint_arch.py:657-706).  x, y: [B,3,H,W] fp32.  With
 * on_device != 0 both are device pointers and the call is asynchronous on
 * `stream`; otherwise they are host pointers and the call copies in, runs and
 * copies out synchronously.  HINT carries `qv_cache` from call to call like
 * the reference module attribute (hint_arch.py:511,660-696); see
 * jirf_net_reset_cache. */
int jirf_net_forward(jirf_net* net, const float* x, float* y, int B, int H, int W, int batch_mode,
                     int on_device, void* stream);

/* HINT's module attribute `self.qv_cache` (hint_arch.py:511): reset = None
 * (cold start, what a freshly constructed model has); get/set copy one
 * [n,C,C] host array (n = 1, or B in per-image mode).  get returns
 * JIRF_ESTATE when the cache is None. */
int jirf_net_reset_cache(jirf_net* net);
int jirf_net_get_cache(jirf_net* net, float* host_out, int* n_out, int* c_out, size_t capacity_floats);
int jirf_net_set_cache(jirf_net* net, const float* host_in, int n, int c);

/* Diagnostics with no counterpart in the reference.  The channel order that
 * ReGroup (hint_arch.py:241-253) derives from an argsort is discontinuous in
 * its input, so two fp32 implementations can legitimately disagree on a near
 * tie; these calls let a checker read the orders the last forward used
 * ([n_attention_blocks][n_groups][C_of_block], int32, block order = execution
 * order) and pin them for the next forwards (NULL clears the pin). */
int jirf_net_attention_blocks(const jirf_net* net, int* n_blocks, int* channels_out, int capacity);
int jirf_net_get_perms(jirf_net* net, int32_t* host_out, size_t capacity_ints, size_t* n_ints);
int jirf_net_set_forced_perms(jirf_net* net, const int32_t* host_in, size_t n_ints);

/* Per-kernel device time of the last forward, measured with HIP events around
 * every launch when enabled (serialises the stream; off by default).
 * names_out receives pointers to static strings. */
int jirf_net_enable_kernel_timing(jirf_net* net, int enable);
int jirf_net_kernel_times(jirf_net* net, const char** names_out, float* ms_out, double* flops_out,
                          double* bytes_out, int* launches_out, int capacity, int* n_out);

/* ---- single-kernel entry points (raw device pointers; used by the known-answer tests and by
 *      rocprofv3 runs).  All tensors fp32 NCHW contiguous unless stated. ---------------------- */

/* y[b,o,p] = sum_i w[o,i] x[b,i,p] (+bias[o]) (+res[b,o,p]) — nn.Conv2d(k=1)
 * (hint_arch.py:156-161,173-178,306,316).  w_host [Cout,Cin], bias_host [Cout] or NULL are HOST
 * arrays (packed and uploaded inside the call); res may be NULL. */
int jirf_op_conv1x1(const float* x, const float* w_host, const float* bias_host, const float* res,
                    float* y, int B, int Cin, int Cout, int P, void* stream);
/* Same with LayerNorm folded in: y = W * LN(x) using per-pixel mean/rstd computed inside the
 * call.  ln_kind 1 = WithBias, 2 = BiasFree (hint_arch.py:98-145). */
int jirf_op_ln_conv1x1(const float* x, const float* ln_w, const float* ln_b, int ln_kind,
                       const float* w_host, float* y, int B, int Cin, int Cout, int P, void* stream);
/* The gated feed-forward branch of a Restormer-style block with its LayerNorm and residual
 * (FeedForward.execute hint_arch.py:150-186 inside TransformerBlock.execute :429-433; the same module in
 * fpro_arch.py:92-114):
 *     y = x + project_out( gelu(d[:hidden]) * d[hidden:] ),  d = dwconv3x3(project_in(LayerNorm(x)))
 * x, y [B,C,H,W] device, distinct buffers.  HOST arrays in the reference's layouts: ln_w, ln_b [C] (ln_kind 1 =
 * WithBias, 2 = BiasFree: ln_b ignored), w_in [2 hidden][C], w_dw [2 hidden][3][3], w_out [C][hidden]; b_in
 * [2 hidden], b_dw [2 hidden], b_out [C] or NULL (the `bias` option).  fused != 0: the one-kernel form, in which
 * the hidden tensor never reaches device memory (C = 96, hidden <= 256, W >= 32; otherwise JIRF_EINVAL); fused == 0:
 * the three launches used for every other shape (LayerNorm-GEMM, depthwise + gate, GEMM + residual).  Synchronises
 * `stream`. */
int jirf_op_gdfn(const float* x, float* y, int B, int C, int hidden, int H, int W, int ln_kind, const float* ln_w_host,
                 const float* ln_b_host, const float* w_in_host, const float* b_in_host, const float* w_dw_host,
                 const float* b_dw_host, const float* w_out_host, const float* b_out_host, int fused, void* stream);
/* No counterpart in the reference.  jirf_op_ln_conv1x1 for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the packed
 * weights and, for Cin > 256, the per-pixel LayerNorm statistics live in a static device array of
 * (JIRF_STATIC_MAX_PACKED_WEIGHT_FLOATS + 2 * JIRF_STATIC_MAX_PIXELS) * 4 bytes = 10 MiB that every GPU
 * the library runs on holds until the process ends; the call makes no hipMalloc / hipFree, copies
 * nothing from the host and does not synchronise (asynchronous on `stream`).  Differences from
 * jirf_op_ln_conv1x1: w_dev [Cout,Cin] is a DEVICE array in the reference layout (packed by a kernel
 * inside the call).  Limits: ceil16(Cin) * ceil128(Cout) <= JIRF_STATIC_MAX_PACKED_WEIGHT_FLOATS and
 * B * P <= JIRF_STATIC_MAX_PIXELS, otherwise JIRF_EINVAL and nothing is launched.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
 * Under stream capture: make ONE call on the device outside any capture first.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress, which may load the code object) and sets a
 * kernel attribute once; neither belongs inside a capture.  Every later call only queues kernels. */
#define JIRF_STATIC_MAX_PACKED_WEIGHT_FLOATS (512 * 1024)
#define JIRF_STATIC_MAX_PIXELS (1024 * 1024)
int jirf_op_ln_conv1x1_static(const float* x, const float* ln_w, const float* ln_b, int ln_kind,
                              const float* w_dev, float* y, int B, int Cin, int Cout, int P, void* stream);
/* Diagnostics, no counterpart in the reference.  The kernels evaluate nn.GELU's exact form
 * 0.5 v (1 + erf(v / sqrt 2)) (hint_arch.py:184,209,226; ast_arch.py:384,456-466; astv2_arch.py:139,184;
 * fpro_arch.py:39-44) through one branch-free polynomial + power of two for erfc instead of the C library's erff.
 * This scans EVERY float in [-hi, hi] on the device and returns maxima to out4_host (host, 4 doubles):
 * [0] |2^-(...) form - erfc(u / sqrt 2)| against float64, [1] |gelu - float64 gelu|, [2] the same for v >= 0 in units of
 * the float spacing at the exact value, [3] |gelu - the erff form|.  Synchronises `stream`. */
int jirf_op_gelu_error_scan(float hi, double* out4_host, void* stream);
/* Arithmetic of the matrix-bound fp32 GEMMs (1x1 and dense kxk convolutions with shared weights; no counterpart in
 * the reference, whose nn.Conv2d is plain fp32).  1 (the default; JIRF_GEMM_SPLIT_BF16=0 in the environment starts
 * with 0): every fp32 operand is written exactly as the sum of three bf16 values and the six largest of the nine
 * partial products, each exact, are accumulated in fp32 on the bf16 matrix cores; the error against a float64
 * product is that of the fp32 matrix instruction or lower (tests/test_split_bf16_gpu.py).  0: v_mfma_f32_32x32x2_f32.
 * The setting is read when a network packs its weights and at every launch: set it before jirf_net_finalize
 * (which packs) and leave it alone while networks packed under the other setting run (one packed with 0 keeps the
 * fp32 instruction whatever the setting).  Returns the previous setting. */
int jirf_set_gemm_split_bf16(int enable);
/* Dense 3x3, stride 1, zero pad 1 (hint_arch.py:441-448,459-466,479-486,648-655).
 * w_host [Cout,Cin,3,3].  mode 0: plain; 1: followed by pixel-unshuffle(2)
 * (hint_arch.py:35-56); 2: followed by pixel-shuffle(2); res (NULL or [B,Cout,H,W]) is
 * added in mode 0. */
int jirf_op_conv3x3(const float* x, const float* w_host, const float* bias_host, const float* res,
                    float* y, int B, int Cin, int Cout, int H, int W, int mode, void* stream);
/* Depthwise 3x3, zero pad 1 (hint_arch.py:163-171,307-315).  w [C,1,3,3] and bias (or NULL) are
 * DEVICE arrays.  gate != 0: C must be even and y[b,c] = gelu(dw(x)[b,c]) * dw(x)[b,c+C/2]
 * for c < C/2 (FeedForward.execute hint_arch.py:182-184). */
int jirf_op_dwconv3x3(const float* x, const float* w, const float* bias, float* y, int B, int C,
                      int H, int W, int gate, void* stream);
/* Per-pixel LayerNorm over channels (hint_arch.py:98-145). kind 1 WithBias, 2 BiasFree. */
int jirf_op_layernorm(const float* x, const float* w, const float* b, float* y, int B, int C, int P,
                      int kind, void* stream);
/* HINT channel attention core (Attention.execute hint_arch.py:351-399 without the two convs and
 * without the intra-modulator): qkv [B,3C,P] device; writes out [B,C,P] (sorted channel order),
 * tcache [B,C,P] (= out + normalised q + normalised k in sorted order, the intra-modulator's
 * input x2 + x1 of hint_arch.py:226,401), perm [G,C] int32, att [B,C,C]
 * (block-diagonal attention after inter-modulation), cache_out [G,C,C].  cache_in NULL = None,
 * else [G,Cc,Cc].  G = B (per-image) or 1 (coupled).  params_host: the reference tensors
 * temperature[4], then for g in 0..3 conv_width.weight[2c,c], conv_width.bias[2c],
 * gating_conv.weight[c,c], gating_conv.bias[c], concatenated (host). */
int jirf_op_hint_attention_core(const float* qkv, const float* params_host, const float* cache_in,
                                int cache_c, float* out, float* tcache, int32_t* perm, float* att,
                                float* cache_out, int B, int C, int P, int batch_mode,
                                void* stream);

/* Head-wise channel attention core of ASTv2 (Attention.execute astv2_arch.py:158-206 without the two
 * convolutions around it: L2-normalised q, k over H W, (q k^T) * temperature[head], softmax branch, the
 * relu^2 -> gelu sparse branch through att_ca (a 1x1 convolution over the ROW axis, weight [2 ch, ch],
 * optional bias [2 ch]), attn0 * (1 + scale) + shift, attention * v) and, with ca_w_host == NULL, of FPro's
 * MDTA (Attention.execute fpro_arch.py:128-144: softmax only).  qkv [B,3C,P] device; writes out [B,C,P]
 * and att [B,C,C] (block-diagonal: head h occupies rows and columns h ch .. (h + 1) ch - 1, zero elsewhere).
 * temperature_host [heads].  norm_add_eps > 0 selects FPro's patched normalize x / (||x|| + eps)
 * (fpro_arch.py:10-12), 0 the x / sqrt(max(sum x^2, tiny)) form. */
int jirf_op_head_attention_core(const float* qkv, const float* temperature_host, const float* ca_w_host,
                                const float* ca_b_host, float norm_add_eps, float* out, float* att, int B, int C,
                                int P, int heads, void* stream);

/* 8x8-window cross attention of FPro's highFrequencyPromptFusion (fpro_arch.py:309-333) between its
 * projections: q [B,C,H,W] (the feature after to_q, unscaled), kv [B,2C,H,W] (the prompt after to_kv: k in the
 * first C channels, v in the rest), channel = head * (C / heads) + d.  Per window and head:
 * softmax((q * d^-0.5) k^T) v, then the reference's transpose to [.., d, N] followed by a RESHAPE to [N, C]
 * (:331-332), written back in window order: out [B,C,H,W] is what project_out consumes.  H, W multiples of 8. */
int jirf_op_cross_window_attention(const float* q, const float* kv, float* out, int B, int C, int heads, int H,
                                   int W, void* stream);

/* FPro's frequency split, dynamic_filter_channel.execute (fpro_arch.py:355-375): global average pool ->
 * 1x1 conv (C -> 9 groups) -> sigmoid gate -> BatchNorm (running statistics) -> softmax over the 9 taps per
 * group -> per-image, per-group 3x3 low-pass with reflection padding; high = x - low.  x, low, high
 * [B,C,H,W] device.  params_host: conv.weight [9 groups, C], conv_gate.weight [9 groups, 9 groups],
 * bn.weight, bn.bias, bn.running_mean, bn.running_var [9 groups] each, concatenated (host). */
int jirf_op_dynamic_filter(const float* x, const float* params_host, int groups, float* low, float* high, int B,
                           int C, int H, int W, void* stream);

/* nn.interpolate(x, size=(Ho, Wo), mode="bilinear" | "bicubic", align_corners=False) as FPro's prompt generator
 * calls it (fpro_arch.py:405,414,425-436; half-pixel centres, edge clamp, bicubic with A = -0.75 as torch): x
 * [B,C,Hi,Wi], y [B,C,Ho,Wo] device. */
int jirf_op_resize(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, int bicubic, void* stream);

/* The high-frequency prompt of frequenctSpecificPromptGenetator (fpro_arch.py:404-411) after its resizes: f
 * [B,C,H,W] (the high part at the level's size), wmap [C,H,W] (the learned weight at that size), body.0
 * (depthwise 1x3: k1 [C,3], b1 [C]) then body.1 (depthwise 3x1: k2 [C,3], b2 [C]), zero padded;
 * y = gelu(body(f)) * f * wmap.  All pointers device. */
int jirf_op_fspg_high(const float* f, const float* k1, const float* b1, const float* k2, const float* b2,
                      const float* wmap, float* y, int B, int C, int H, int W, void* stream);

/* ---- script-level image plumbing of the test scripts, on the device (all pointers are device
 *      pointers except the start lists; index / byte work, bit-exact) --------------------------- */

/* nn.pad(input_, (0, pad_right, 0, pad_bottom), "reflect") of Deraining/test_rain100L.py:147-152
 * (same lines in every <Task>/test_*.py): y [B,C,H+pad_bottom,W+pad_right]. */
int jirf_op_reflect_pad(const float* x, float* y, int B, int C, int H, int W, int pad_bottom, int pad_right,
                        void* stream);
/* splitimage (test_rain100L.py:31-48): tiles [(nh*nw)*B, C, crop, crop], tile t = (row start
 * t / nw, column start t % nw) as in the reference's list order; hstarts / wstarts are HOST int32
 * arrays (at most 96 entries each). */
int jirf_op_split_tiles(const float* x, float* tiles, int B, int C, int H, int W, int crop, const int32_t* hstarts,
                        int nh, const int32_t* wstarts, int nw, void* stream);
/* mergeimage with the uniform score map the scripts use (test_rain100L.py:51-75, is_mean=True):
 * out [B,C,H,W] = sum of the covering tiles in list order / their count.  The tiles must cover
 * the image. */
int jirf_op_merge_tiles(const float* tiles, float* out, int B, int C, int H, int W, int crop, const int32_t* hstarts,
                        int nh, const int32_t* wstarts, int nw, void* stream);
/* restored[:, :, :h, :w] clamped to [0, 1] (test_rain100L.py:171-175): y [B,C,h,w]. */
int jirf_op_crop_clamp(const float* x, float* y, int B, int C, int H, int W, int h, int w, void* stream);
/* The same followed by .permute(0,2,3,1) and skimage img_as_ubyte (test_rain100L.py:176-190):
 * y [B,h,w,C] uint8 = rint(clamp(x, 0, 1) * 255) (float32 product, round half to even). */
int jirf_op_crop_clamp_u8(const float* x, uint8_t* y, int B, int C, int H, int W, int h, int w, void* stream);
/* calculate_psnr (basicsr/metrics/psnr_ssim.py:11-69) on two planar [C,H,W] device images, as its three
 * ingredients: after the call completes on `stream`, work[0] = sum of squared differences (float64),
 * work[1] = maximum of the first image (psnr_ssim.py:68 picks the peak value from it), work[2] = number of
 * values compared.  crop_border as :57-59.  quantise != 0 first makes the uint8 images of tensor2img
 * (basicsr/utils/img_util.py:136-166: clamp [0,1], * 255, round half to even) - the `use_image: true`
 * path of image_restoration_model.py:389-394.  y_channel != 0 (needs C == 3 and values in [0,255], i.e.
 * quantise or 8-bit input) compares the BT.601 luma of to_y_channel (basicsr/metrics/metric_util.py:34-47);
 * blue_plane = 2 when planes are R,G,B and the caller had rgb2bgr on, 0 when it was off.
 * `work` is DEVICE memory of jirf_op_psnr_work_doubles() float64 values, owned by the caller. */
int jirf_op_psnr_work_doubles(void);
int jirf_op_psnr_stats(const float* a, const float* b, int C, int H, int W, int crop_border, int quantise, int y_channel,
                       int blue_plane, double* work, void* stream);

/* calculate_ssim (basicsr/metrics/psnr_ssim.py:247-330) on two planar [C,H,W] device images (C <= 4): after the call
 * completes on `stream`, work[0] = sum of the SSIM map, work[1] = number of map values (the metric is their ratio).
 * Colour form (_ssim_3d, :161-199): one 11 x 11 x 11 Gaussian window (sigma 1.5) over the H x W x C volume with
 * replicate padding, i.e. also across the colour planes; y_channel != 0: the 2-D window of _ssim_cly (:202-244) on the
 * BT.601 luma.  max_value is the reference's `1 if img1.max() <= 1 else 255` (:317; 255 for the luma form) and sets
 * C1 = (0.01 max)^2, C2 = (0.03 max)^2.  crop_border, quantise, y_channel, blue_plane as jirf_op_psnr_stats.
 * `work` is DEVICE memory of jirf_op_ssim_work_doubles(C, H, W) float64 values, owned by the caller. */
int jirf_op_ssim_work_doubles(int C, int H, int W);
int jirf_op_ssim_stats(const float* a, const float* b, int C, int H, int W, int crop_border, int quantise, int y_channel,
                       int blue_plane, float max_value, double* work, void* stream);

/* ---- training step, device side: the operators of ImageCleanModel.optimize_parameters
 *      (basicsr/models/image_restoration_model.py:235-276) for AST - forward operators that read the weights from
 *      DEVICE memory in the reference's layouts (they change every step), their data and weight gradients, the
 *      loss and the optimiser update.  Raw device pointers, fp32, planar [B][C][P].  An activation argument is
 *      followed by its batch stride in floats (0 = dense, C * P), so that a channel slice of a wider tensor is a
 *      pointer offset plus that tensor's stride.  All calls are asynchronous on `stream` unless stated.  The
 *      operators share one scratch area per device (packed weights, partial sums; grown on demand, kept until the
 *      process ends): training calls that overlap on one device race.  Every reduction has a fixed order (no
 *      atomics): a step is reproducible. */

/* Number of times any training scratch buffer of any device has been reallocated since the library was loaded.  A
 * caller that replays a captured step (hipGraph) records it at capture and must capture again once it has changed:
 * the recorded kernels hold the old addresses.  A call that would have to grow a buffer while its stream is being
 * captured fails with JIRF_ESTATE and leaves every buffer as it was.  No counterpart in the reference. */
long jirf_train_scratch_generation(void);

/* Per-kernel device time of the training operators called from this thread (HIP events around every launch;
 * serialises nothing but adds two events per launch).  jirf_train_kernel_times waits for the recorded launches,
 * returns the totals per label since the last call (names_out receives pointers to static strings) and starts
 * a new interval.  No counterpart in the reference. */
int jirf_train_kernel_timing(int enable);
int jirf_train_kernel_times(const char** names_out, float* ms_out, double* flops_out, double* bytes_out, int* launches_out,
                            int capacity, int* n_out);

/* Pixel loss of the shipped configurations: "L1Loss" is nn.smooth_l1_loss with mean reduction
 * (basicsr/models/losses/losses.py:12-14,44-52).  loss = loss_weight * mean(l) goes to *loss_dev (device, or NULL)
 * and to *loss_host (or NULL; a host destination synchronises `stream`, a device destination alone does not, so
 * the caller can keep queueing the backward pass and read the loss later); grad (or NULL) receives
 * d loss / d pred = loss_weight * clamp(pred - target, -1, 1) / n. */
int jirf_train_smooth_l1(const float* pred, const float* target, float* grad, long n, float loss_weight,
                         double* loss_dev, double* loss_host, void* stream);
/* Sum of squares of a flat buffer in double (the norm of Optimizer.clip_grad_norm,
 * image_restoration_model.py:269-270), to device and / or host memory (host: synchronises `stream`). */
int jirf_train_sumsq(const float* x, long n, double* out_dev, double* out_host, void* stream);
/* Gradient scale + clip + one Adam step on flat buffers p, g, m, v of n floats.  The gradient used is
 * grad_scale * g (grad_scale = 1 / ranks after a SUM all-reduce of g: the data-parallel mean Jittor's optimizer
 * takes under MPI, without a pass of its own over the buffer; 1 otherwise), clipped when max_norm > 0 by
 * min(max_norm / (grad_scale * sqrt(*sumsq_dev) + 1e-6), 1) with *sumsq_dev the sum of squares of g as stored.
 * step counts the optimiser's own updates from 1 (Adam's bias correction).  torch_form 0: Jittor's update
 * p -= m * lr * sqrt(1 - b1^t) / (1 - b0^t) / (sqrt(v) + eps) (what optimizer_g.step() runs,
 * image_restoration_model.py:215,271); 1: torch.optim.Adam's placement of eps. */
int jirf_train_adam_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq_dev, float max_norm,
                         float grad_scale, float lr, float beta0, float beta1, float eps, int step, int torch_form,
                         void* stream);

/* Convolution / Linear on device weights, as an implicit GEMM (the forward kernels of the inference path;
 * the weights are packed by a kernel inside the call).  w_kind selects how `w` is read:
 *   JIRF_W_FWD            w [Cout][Cin][ks][ks] (nn.Conv2d; ks = 1: nn.Linear [Cout][Cin]), geometry ks/stride/pad:
 *                         y [B,Cout,Hout,Wout] from x [B,Cin,Hin,Win]  (ast_arch.py:100-101,456-474,558-565,630-636)
 *   JIRF_W_BWD_DATA       the data gradient of a stride-1 convolution whose FORWARD weight is w [Cin][Cout][ks][ks]
 *                         and forward padding `pad`: x = dY [B,Cin,..], y = dX [B,Cout,..]   (ks = 1: W^T dY)
 *   JIRF_W_DECONV         nn.ConvTranspose2d(Cin, Cout, 2, stride 2) (ast_arch.py:653-660): w [Cin][Cout][2][2],
 *                         y [B,Cout,2 Hin,2 Win]; ks/stride/pad are ignored.  Its data gradient is JIRF_W_FWD with
 *                         the same w read as [Cout' = Cin][Cin' = Cout][2][2], ks 2, stride 2, pad 0.
 *   JIRF_W_DOWN_BWD_DATA  the data gradient of nn.Conv2d(k 4, stride 2, pad 1) (ast_arch.py:630-636) with forward
 *                         weight w [Cin][Cout][4][4]: x = dY [B,Cin,Hin,Win], y = dX [B,Cout,2 Hin,2 Win]
 * epi: 0 y = acc + bias; 1 y = acc + bias + r; 6 y = gelu(acc + bias) (ks = 1); 7 LeakyReLU(0.01) (ks > 1).
 * bias [Cout] or NULL (device).  r / r_bs only with epi 1. */
#define JIRF_W_FWD 0
#define JIRF_W_BWD_DATA 1
#define JIRF_W_DECONV 2
#define JIRF_W_DOWN_BWD_DATA 4
#define JIRF_W_PACKED 8 /* or-ed into w_kind: `w` is that kind's packed form, written by jirf_train_pack_plan_run */
int jirf_train_conv(const float* x, long x_bs, const float* w, int w_kind, const float* bias, const float* r, long r_bs,
                    float* y, long y_bs, int B, int Cin, int Cout, int Hin, int Win, int ks, int stride, int pad,
                    int epi, void* stream);
/* The packed forms of many weights in one launch.  Without a plan every jirf_train_conv call packs its weight
 * by a kernel of its own (a few hundred small launches per AST step); a plan lists the (weight, kind) pairs of a
 * step once - w[i] (device, reference layout), bias[i] (JIRF_W_DECONV only, or NULL), w_kind[i], Cin[i], Cout[i],
 * ks[i] as they are passed to jirf_train_conv, and wt[i], a device buffer of jirf_train_packed_floats() floats
 * that receives the packed form - and jirf_train_pack_plan_run() rewrites every wt[i] from the current weights
 * (call it after each update of the weights).  jirf_train_conv then takes wt[i] with w_kind | JIRF_W_PACKED
 * (and still the bias pointer).  The plan belongs to the device current at creation.  No counterpart in the
 * reference, whose framework keeps its own weight layouts. */
long jirf_train_packed_floats(int w_kind, int Cin, int Cout, int ks);   /* < 0: bad arguments */
int jirf_train_pack_plan_create(int n, const float* const* w, const float* const* bias, float* const* wt,
                                const int* w_kind, const int* Cin, const int* Cout, const int* ks, void** plan_out);
int jirf_train_pack_plan_run(void* plan, void* stream);
void jirf_train_pack_plan_destroy(void* plan);
/* Weight and bias gradient of a convolution / Linear: dw [M][Cin][ks][ks] (+)= sum_{b,oy,ox} dy[b][o][oy][ox]
 * x[b][c][oy stride - pad + ky][ox stride - pad + kx], db [M] (+)= sum dy (db may be NULL).  Hout * Wout must be
 * a multiple of 16.  The weight gradient of the transposed convolution is this call with the roles swapped
 * (dy := its input, x := its output gradient, ks 2, stride 2, pad 0), which yields [Cin][Cout][2][2]. */
int jirf_train_conv_bwd_weight(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, float* db, int B,
                               int M, int Cin, int Hout, int Wout, int Hin, int Win, int ks, int stride, int pad,
                               int accumulate, void* stream);
/* The same for dense [B][M][P] / [B][K][P] operands of a Linear (P % 4 == 0). */
int jirf_train_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int B, int M, int K, int P,
                                 int accumulate, void* stream);
/* Per-pixel LayerNorm over channels with device weights (ast_arch.py:694,718,743 nn.LayerNorm = kind 1; kind 2 = the
 * BiasFree form of hint_arch.py:98-117), and its backward: dx [B,C,P] (+ add, the gradient arriving over the
 * shortcut around the normalised branch, or NULL), dw [C], db [C] (kind 1) from x, dy and the weight w. */
int jirf_train_layernorm(const float* x, long x_bs, const float* w, const float* b, float* y, long y_bs, int B, int C,
                         int P, int kind, void* stream);
int jirf_train_layernorm_bwd(const float* x, long x_bs, const float* dy, long dy_bs, const float* w, const float* add,
                             long add_bs, float* dx, long dx_bs, float* dw, float* db, int B, int C, int P, int kind,
                             int accumulate, void* stream);
/* Elementwise pieces; one image = n_img contiguous floats.  Forward mode 0: y = gelu(pre) (ast_arch.py:384,456);
 * mode 1: y = gelu(pre) * x2 (the FRFN gate, ast_arch.py:507-510); mode 3: y = gelu(pre) * gelu(x2) (the same gate on
 * linear1's raw second half, so that the GELU after linear1 is a pass over the first half only).  Backward mode 0:
 * dxa = dy * gelu'(pre); mode 1: dxa = dy * x2 * gelu'(pre), dxb = dy * gelu(pre); mode 3: dxa = dy * gelu(x2) *
 * gelu'(pre), dxb = dy * gelu(pre) * gelu'(x2); mode 2: LeakyReLU(0.01) with pre = the activation's
 * output: dxa = dy * (pre > 0 ? 1 : 0.01). */
int jirf_train_gelu(const float* pre, long pre_bs, const float* x2, long x2_bs, float* y, long y_bs, int B, long n_img,
                    int mode, void* stream);
int jirf_train_gelu_bwd(const float* pre, long pre_bs, const float* x2, long x2_bs, const float* dy, long dy_bs,
                        float* dxa, long dxa_bs, float* dxb, long dxb_bs, int B, long n_img, int mode, void* stream);
/* y = a (+ b, or NULL): concatenation copies and the sums of gradients that meet at a shortcut. */
int jirf_train_add(const float* a, long a_bs, const float* b, long b_bs, float* y, long y_bs, int B, long n_img,
                   void* stream);
/* y = a + scale[image] * b (a may be NULL: y = scale[image] * b); scale [B] is a device array.  A branch under
 * stochastic depth (DropPath, ast_arch.py:52-70,837,840): its output, and in the backward its output gradient, is
 * multiplied per sample by 0 or 1 / keep_prob. */
int jirf_train_scale_add(const float* a, long a_bs, const float* b, long b_bs, const float* scale, float* y, long y_bs,
                         int B, long n_img, void* stream);
/* The per-sample DropPath factors of a step, drawn on the device: out [rows][B], out[r][b] = 0 with probability
 * rate[r] and 1 / (1 - rate[r]) otherwise (drop_path(), ast_arch.py:62-70); rate [rows] and counter [1] are device
 * arrays.  The variates are a hash of (seed, *counter, r, b) and the call advances *counter, so the sequence is a
 * function of the seed (the reference draws from Jittor's generator, whose stream is not reproduced). */
int jirf_train_drop_path_scales(float* out, const float* rate, int rows, int B, unsigned long long seed,
                                unsigned long long* counter, void* stream);
/* out [C] (+)= sum_{b,p} x[b][c][p] (the bias gradient of the transposed convolution). */
int jirf_train_channel_sum(const float* x, long x_bs, float* out, int B, int C, long P, int accumulate, void* stream);
/* Depthwise 3x3 (zero pad 1, ast_arch.py:458-466) on device weights w [C,1,3,3], bias [C] or NULL; flip != 0 reverses
 * the taps (the data gradient: x = dY, bias NULL).  Weight / bias gradient: dw [C,1,3,3], db [C] (or NULL). */
int jirf_train_dwconv3x3(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs, int B, int C,
                         int H, int W, int flip, void* stream);
int jirf_train_dwconv3x3_bwd_weight(const float* x, long x_bs, const float* dy, long dy_bs, float* dw, float* db, int B,
                                    int C, int H, int W, int accumulate, void* stream);
/* AST's sparse window attention (WindowAttention_sparse.execute ast_arch.py:275-318 with the roll / partition /
 * mask / reverse of :786-837 as addressing) on qkv [B,3C,H*W] (q | k | v channel blocks, before the q scaling),
 * 8 x 8 windows, head dim 32, shift 0 or 4; table [225][heads] and w2 [2] (the raw mixing weights `w`) are device
 * arrays.  coupled 0: the reference's (window, head) pairing of k runs per image (B separate B = 1 calls);
 * 1: over the whole batch (the literal B > 1 call, what a training batch does).  Backward: dqkv [B,3C,H*W],
 * dtable [225][heads], dw2 [2]. */
int jirf_train_window_attention(const float* qkv, long qkv_bs, const float* table, const float* w2, float* out,
                                long out_bs, int B, int C, int H, int W, int heads, int shift, int coupled, void* stream);
int jirf_train_window_attention_bwd(const float* qkv, long qkv_bs, const float* table, const float* w2, const float* dout,
                                    long dout_bs, float* dqkv, long dqkv_bs, float* dtable, float* dw2, int B, int C, int H,
                                    int W, int heads, int shift, int coupled, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JIRF_HIP_H */
