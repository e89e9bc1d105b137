// Shared host/device declarations of libjirf_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/jirf_hip.h"

namespace jirf {

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* last_error();

struct Failure {
  int code;
};

#define JIRF_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      jirf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                      __LINE__);                                                          \
      throw jirf::Failure{JIRF_EHIP};                                                     \
    }                                                                                     \
  } while (0)

#define JIRF_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      jirf::set_error(__VA_ARGS__);    \
      throw jirf::Failure{code};       \
    }                                  \
  } while (0)

// ---- tensor views ------------------------------------------------------------------------
// A planar activation: element (b, c, p) lives at ptr[b * bs + c * P + p]; a view may be a
// channel slice of a wider buffer (concat-free skips), hence the explicit batch stride.
struct TV {
  float* p = nullptr;
  long bs = 0;
};
inline TV tv(float* p, long bs) { return TV{p, bs}; }
inline TV tv_chan(TV t, long c, long P) { return TV{t.p + c * P, t.bs}; }

// ---- GEMM family ---------------------------------------------------------------------------
// Y[b][o][p] = epi( sum_k Wt[b][k][o] * Xk[b][k][p] ).  Wt is the K-major packed weight
// (rows padded to GEMM_KPAD with zeros, row length Mpad padded to GEMM_MPAD).
constexpr int GEMM_BK = 16;
constexpr int GEMM_KPAD = 16;
constexpr int GEMM_MPAD = 128;

enum GemmLoader : int {
  LD_PLAIN = 0,   // X[b][k][p]
  LD_GATHER = 1,  // X[b][kidx[g][k]][p]
  LD_LN = 2,      // (X[b][k][p] - mu[b][p]) * rstd[b][p] * lnw[k] + lnb[k]
  LD_CONV3 = 3,   // implicit k x k conv: k = tap * Cin + c, X[b][c][y*stride-pad+ky][x*stride-pad+kx],
                  // zero padded (3x3 s1 p1 and 4x4 s2 p1 are the shapes the networks use)
};

enum GemmEpi : int {
  EP_STORE = 0,      // y = acc + bias
  EP_RESID = 1,      // y = acc + bias + R
  EP_GELU_MUL = 2,   // y = gelu(acc + bias) * AUX
  EP_AV_INTRA = 3,   // y = acc; y2 = acc + q[perm[o]] * rq[perm[o]] + k[perm[o]] * rk[perm[o]]
  EP_UNSHUFFLE = 4,  // conv3x3 then pixel-unshuffle(2)
  EP_SHUFFLE = 5,    // then pixel-shuffle(2) (also the store of a 2x2 stride-2 transposed conv)
  EP_GELU = 6,       // y = gelu(acc + bias)
  EP_LEAKY = 7,      // y = leaky_relu(acc + bias, 0.01)
};

struct GemmArgs {
  // problem
  int B, M, K, P;        // M = Cout, K = reduction length (Cin or 9*Cin), P = pixels per image
  int Mpad;              // packed row length of Wt
  const float* wt;       // [Kpad][Mpad]
  long wt_bs;            // batch stride of wt (0: shared)
  const uint16_t* wt3 = nullptr;  // optional: the same weights split into three bf16 planes (see pack_wt_split3)
  long wt3_plane = 0;             // elements per plane
  long wt3_bs = 0;                // batch stride of the planes in elements (0: shared)
  const float* bias;     // [M] or null
  long bias_bs = 0;      // > 0: one bias vector per image, bias + b * bias_bs
  TV x;                  // input activation
  TV y;                  // output
  // loader extras
  const int* kidx;       // LD_GATHER: [G][K] row indices
  int imgs_per_group;    // LD_GATHER / EP_AV_INTRA: images sharing one index row
  const float* mu;       // LD_LN: [B][P]
  const float* rstd;     // LD_LN: [B][P]
  const float* lnw;      // LD_LN: [K]
  const float* lnb;      // LD_LN: [K] or null (BiasFree: mu is also ignored)
  int ln_center;         // LD_LN: subtract mu (WithBias) or not (BiasFree)
  int H, W, Cin;         // LD_CONV3 geometry: OUTPUT height/width (H * W == P) and input channels
  int ks, stride, pad;   // LD_CONV3: kernel edge, stride, zero padding (0 0 0 means 3 1 1)
  int Hin, Win;          // LD_CONV3: input plane size (0 0 means H W)
  TV x2;                 // LD_PLAIN: rows k < k_split are read from x2 instead of x
  int k_split;
  // epilogue extras
  int epi;
  TV r;                  // EP_RESID residual / EP_GELU_MUL aux
  TV y2;                 // EP_AV_INTRA second output
  TV q, k;               // EP_AV_INTRA sources (unsorted q, k planes)
  const float* rq;       // EP_AV_INTRA: [B][C] reciprocal norms of q
  const float* rk;       // EP_AV_INTRA: [B][C]
  // split-K workspace (optional): when the grid of the K-loop kernel would leave most CUs idle, the
  // reduction is cut into slabs that are written here ([split][B][M][P]) and summed, in slab order, by a
  // second kernel that also applies the epilogue.  Null: never split.
  float* splitk_ws = nullptr;
  size_t splitk_floats = 0;
};
constexpr size_t SPLITK_WS_FLOATS = (size_t)4 << 20;   // what the networks reserve in their arena (16 MiB)

void launch_gemm(const GemmArgs& a, int loader, hipStream_t s);
// true when the LD_LN loader computes the LayerNorm statistics itself (mu / rstd may be null)
bool gemm_ln_inline(int K);
// the same for one launch shape (M outputs, P pixels, B images): some shapes go to a kernel that reads the statistics
bool gemm_ln_inline(int K, int M, int P, int B);
// true when the split-bf16 GEMM variant is switched on (JIRF_GEMM_SPLIT_BF16=1): weights are then
// also packed as three bf16 planes
bool gemm_split_bf16_enabled();
// reduction lengths the split-bf16 kernel is built for (the packers make the bf16 planes for these)
bool gemm_split_bf16_covers(int K);
// w [M][K] fp32 -> three bf16 planes (hi, mid, lo with hi + mid + lo == w) laid out
// [plane][Kp / 8][Mpad][8]: the 8 consecutive k of one MFMA operand are one 16-byte word per row.
// Returns elements per plane.
size_t pack_wt_split3(const float* w, int M, int K, int Mpad, std::vector<uint16_t>& out);
// the same from a K-major packed matrix [Kp][Mpad] (any of the packers below)
size_t pack_split3_kmajor(const std::vector<float>& packed, int Kp, int Mpad, std::vector<uint16_t>& out);
// split-bf16 GEMM with the pixel operand in registers (gemm_split.hip): whether it serves this launch, and the launch
bool gemm_b3_covers(const GemmArgs& a, int loader);
void launch_gemm_b3(const GemmArgs& a, int loader, hipStream_t s);
// per-image K-major fp32 matrices [B][Kp][Mpad] -> three bf16 planes each, [B][3][Kp / 8][Mpad][8] (on the device)
void launch_split_planes(const float* w, uint16_t* planes, int B, int Kp, int Mpad, hipStream_t s);
// ... and its form for short outputs (M <= 96), the reduction walked in chunks
bool gemm_k3_covers(const GemmArgs& a, int loader);
void launch_gemm_k3(const GemmArgs& a, int loader, hipStream_t s);
// host-side packers (return Kpad; out is resized to Kpad * Mpad)
int pack_wt_1x1(const float* w, int M, int K, std::vector<float>& out, int* Mpad);
int pack_wt_3x3(const float* w, int M, int Cin, std::vector<float>& out, int* Mpad);
int pack_wt_conv(const float* w, int M, int Cin, int ks, std::vector<float>& out, int* Mpad);

// ---- memory-bound kernels -------------------------------------------------------------------
void launch_layernorm(TV x, const float* w, const float* b, TV y, int B, int C, int P, int kind,
                      hipStream_t s);
void launch_ln_stats(TV x, float* mu, float* rstd, int B, int C, int P, hipStream_t s);
// ---- FPro frequency-prompt operators (fpro_ops.hip) ------------------------------------------------
struct DynFilterW {            // dynamic_filter_channel parameters, device pointers
  const float *conv, *gate, *bn_w, *bn_b, *bn_mean, *bn_var;
  int groups;
};
void launch_channel_mean(TV x, float* out, int B, int C, long P, hipStream_t s);
// pooled [B][C], coeff [B][groups*9] are scratch; low / high [B][C][H*W]
void launch_dynfilter(TV x, const DynFilterW& w, float* pooled, float* coeff, TV low, TV high, int B, int C,
                      int H, int W, hipStream_t s);
// x: planes at x + b * x_bs + c * x_cs (x_bs = 0 resizes one parameter map for every image)
void launch_resize(const float* x, long x_bs, long x_cs, float* y, long y_bs, int B, int C, int Hi, int Wi, int Ho,
                   int Wo, bool cubic, hipStream_t s);
void launch_fspg_high(TV f, const float* k1, const float* b1, const float* k2, const float* b2, const float* wmap,
                      TV y, int B, int C, int H, int W, hipStream_t s);
void launch_cross_window_attention(TV q, TV kv, TV out, int B, int C, int heads, int H, int W, hipStream_t s);
void launch_prompt_low(const float* mean, float hw, const float* Wb, const float* bb, const float* w00,
                       float ifft_scale, const float* M, const float* b0, float* bias_out, int B, int C2, int C,
                       hipStream_t s);

// ---- script-level image plumbing (harness.hip) ---------------------------------------------------
constexpr int TILE_GRID_MAX = 96;
struct TileGrid {            // tiles = row starts x column starts, all of one square size
  int crop, nh, nw;
  int hs[TILE_GRID_MAX], ws[TILE_GRID_MAX];
};
void launch_reflect_pad(const float* x, float* y, int B, int C, int H, int W, int pad_bottom, int pad_right,
                        hipStream_t s);
void launch_split_tiles(const float* x, float* tiles, const TileGrid& g, int B, int C, int H, int W, hipStream_t s);
void launch_merge_tiles(const float* tiles, float* out, const TileGrid& g, int B, int C, int H, int W, hipStream_t s);
void launch_crop_clamp(const float* x, void* y, int B, int C, int H, int W, int h, int w, bool to_u8, hipStream_t s);
size_t psnr_work_doubles();
void launch_psnr_stats(const float* a, const float* b, int C, int H, int W, int border, int quantise, int y_channel,
                       int blue_plane, double* work, hipStream_t s);
size_t ssim_work_doubles(int C, int H, int W);
void launch_ssim_stats(const float* a, const float* b, int C, int H, int W, int border, int quantise, int y_channel,
                       int blue_plane, float max_value, double* work, hipStream_t s);

// Dense 3x3 (stride 1, zero pad 1) with at most 4 output channels, on the vector ALUs; weights in
// the GEMM's K-major packing wt[(tap * Cin + c) * Mpad + o].  Returns false (nothing launched)
// when the shape is not one it is built for.
bool try_launch_conv3x3_fewout(const struct GemmArgs& a, hipStream_t s);
// gate: 0 plain, 1 gelu(dw(x1)) * dw(x2), 2 dw(x1) * x2, 3 gelu(dw(x1)) * x2, 4 gelu(dw(x)).  wmod/wconv: ASTv2 partial convolution
// (channel c uses weight c % wmod when c % wmod < wconv, and is copied through otherwise).
void launch_dwconv3x3(TV x, const float* w, const float* bias, TV y, int B, int C, int H, int W,
                      int gate, hipStream_t s, int wmod = 0, int wconv = 0);

// Accuracy of the branch-free GELU over every float in [-hi, hi] (see gelu_scan_kernel); synchronises `s`.
void gelu_error_scan(float hi, double out[4], hipStream_t s);

// ---- HINT channel attention ------------------------------------------------------------------
struct HintAttnShape {
  int B, C, P;
  int G;               // statistic groups: B (per image) or 1 (coupled)
  int gsz[4], goff[4]; // channel-group sizes / offsets in sorted order
  int splits;          // pixel splits of the statistics kernel
};
HintAttnShape hint_attn_shape(int B, int C, int P, int batch_mode);

struct HintAttnWeights {   // device pointers
  const float* temperature;             // [4]
  const float* cw_wt[4];                // conv_width.weight packed K-major by pack_hint_modulator: [c][2c -> x32]
  const float* cw_b[4];                 // [2c]
  const float* gc_wt[4];                // gating_conv.weight packed K-major: [c][c -> x32]
  const float* gc_b[4];                 // [c]
};
// w [M][K] (a Conv1d(k=1) kernel) -> [K][M rounded up to 32], zero padded
void pack_hint_modulator(const float* w, int M, int K, std::vector<float>& out);

struct HintAttnScratch {   // device scratch, carved by hint_attn_scratch_carve
  double* gram_part;   // [G][splits][C][C]
  float* qk_part;      // [B][splits][C][C]
  double* qsum_part;   // [G][splits][C]
  float* q2_part;      // [B][splits][C]
  float* k2_part;      // [B][splits][C]
  float* qmean;        // coupled mode only: [C][P] batch mean of q
  double* sim;         // [G][C] mean similarity
  float* s_qk;         // [B][C][C]
  float* rq;           // [B][C] reciprocal L2 norms of the q rows
  float* rk;           // [B][C]
  float* att_t;        // [B][Kpad][Mpad] block-diagonal attention, transposed (GEMM weight layout)
  uint16_t* att3;      // [B][3][Kpad / 8][Mpad][8] att_t as three bf16 planes (filled when the split GEMMs are on)
  float* att_nat;      // [B][C][C] block-diagonal attention, natural orientation
  float* pooled;       // [G][C][C] carried cache pooled to this block's C
  float* pooled_g;     // [G][sum c_g^2] the same pooled once more to every group's c x c
};
// Lays the scratch out from `base` (may be null to size it) and returns the bytes needed.
size_t hint_attn_scratch_carve(const HintAttnShape& sh, void* base, HintAttnScratch* out);

// Runs regroup statistics -> permutation -> per-group softmax -> inter-modulation -> cache
// update.  q and k are channel slices of the dw-conv output.  perm_out [G][C];
// forced_perm (or null) overrides the computed order.  cache_in null = None.
void launch_hint_attention(const HintAttnShape& sh, TV q, TV k, const HintAttnWeights& w,
                           const float* cache_in, int cache_c, float* cache_out,
                           const HintAttnScratch& sc, int* perm_out, const int* forced_perm,
                           hipStream_t s);

// ---- head-wise channel attention (ASTv2) -------------------------------------------------------
struct HeadAttnScratch {
  float* qk_part;  // [B][splits][C][C]
  float* q2_part;  // [B][splits][C]
  float* k2_part;  // [B][splits][C]
  float* s_qk;     // [B][C][C]
  float* rq;       // [B][C]
  float* rk;       // [B][C]
  float* att_t;    // [B][Kpad][Mpad] block-diagonal attention, transposed (GEMM weight layout)
  uint16_t* att3;  // [B][3][Kpad / 8][Mpad][8] the same as three bf16 planes (filled by the caller when it wants them)
};
int head_attn_splits(int B, int C, int P);
size_t head_attn_scratch_carve(int B, int C, int P, int splits, void* base, HeadAttnScratch* out);
// q, k: channel slices of the dw-conv output.  Leaves the per-image attention in sc.att_t.
// ca_w == NULL: plain MDTA (softmax only, fpro_arch.py:128-144); otherwise ASTv2's sparse branch and
// att_ca.  norm_add_eps > 0: rows are normalised as x / (||x|| + eps) (fpro_arch.py:10-12) instead
// of x / max(||x||, tiny).
void launch_head_attention(int B, int C, int P, int heads, TV q, TV k, const float* temperature,
                           const float* ca_w, const float* ca_b, const HeadAttnScratch& sc,
                           hipStream_t s, float norm_add_eps = 0.f);

// ---- 8x8 window attention (AST) ---------------------------------------------------------------
struct WinAttnArgs {
  int B, C, H, W, heads;     // head_dim = C / heads must be 32, window 8 x 8
  int shift;                 // 0 or 4: cyclic shift of the window grid (and the shift mask)
  int coupled;               // 0: the reference's K (window, head) remap runs per image (B = 1 calls);
                             // 1: over the whole batch (a literal B > 1 call)
  TV q, k, v;                // planar [C][H*W] slices
  TV out;                    // planar [C][H*W]
  const float* table;        // relative_position_bias_table [225][heads]
  float scale;               // head_dim^-0.5
  float w_softmax, w_relu2;  // softmax([w0, w1]) mixing weights
  const float* w_dev = nullptr;  // training: the raw [w0, w1] in device memory (they change every step); overrides the two above
};
void launch_window_attention(const WinAttnArgs& a, hipStream_t s);

// exp(x) for the softmax kernels: one v_exp_f32 on x * log2(e).  The product's rounding adds |x| * 6e-8 to the
// relative error (arguments here are score - row maximum, in [-100, 0]); arguments below -87 give 0 or a denormal,
// as expf does.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// ---- kernel accounting -----------------------------------------------------------------------
struct KernelClock;
KernelClock* current_clock();
void set_current_clock(KernelClock* c);
// RAII marker placed around a launch: records (name, flops, bytes) and, when a clock is
// installed, brackets the launch with HIP events.
// Returns a pointer that stays valid for the life of the process (labels carry shapes).
const char* intern_label(const char* fmt, ...);
#ifdef __HIPCC__
// ---- GELU (exact erf form, 0.5 v (1 + erf(v / sqrt 2)): nn.GELU of the reference's framework) -------------------
// Branch-free, eleven vector instructions, one of them transcendental (the C library's erff costs two code paths
// whenever the lanes of a wave straddle |x| = 1, which activations do).  With u = |v|,
//     half_erfc(u) = 0.5 erfc(u / sqrt 2) = 2^-(1 + u Q(u)),     gelu(v) = max(v, 0) - u half_erfc(u).
// -log2 erfc(u / sqrt 2) is smooth and nearly linear-to-quadratic on the whole axis (erfc has no real zero), so ONE
// polynomial u Q(u) of degree 8 serves every u: the coefficients are the minimax fit on [0, 6] of the error weighted
// by erfc itself (what reaches the result), 1.6e-8 before rounding.  Past 6 the polynomial keeps rising (leading
// coefficient > 0), the power of two underflows to 0 and gelu(v) = max(v, 0) exactly, as erfc < 2^-30 there asks.
// Measured over every float in [-8.5, 8.5] (jirf_op_gelu_error_scan): |2 half_erfc - erfc| <= 2e-7 and gelu within
// 2 ulp of the float64 value.  NaN stays NaN; +inf gives NaN (inf * 0) where the erf form gives inf.
__device__ __forceinline__ float half_erfc_abs(float u) {
  float r = fmaf(u, 2.834913e-06f, -3.937769e-05f);
  r = fmaf(r, u, 0.00018618023f);
  r = fmaf(r, u, 0.0001369364f);
  r = fmaf(r, u, -0.00706342f);
  r = fmaf(r, u, 0.05249618f);
  r = fmaf(r, u, 0.4592082f);
  r = fmaf(r, u, 1.1511052f);
  return __builtin_amdgcn_exp2f(-fmaf(r, u, 1.0f));      // v_exp_f32
}
__device__ __forceinline__ float gelu_fast(float v) {
  const float u = fabsf(v);
  return fmaf(-u, half_erfc_abs(u), fmaxf(v, 0.0f));
}
// d gelu / d v = Phi(v) + v phi(v), Phi(v) = v >= 0 ? 1 - half_erfc(|v|) : half_erfc(|v|)
__device__ __forceinline__ float gelu_grad_fast(float v) {
  const float h = half_erfc_abs(fabsf(v));
  const float cdf = v >= 0.0f ? 1.0f - h : h;
  // exp(-v^2 / 2) / sqrt(2 pi) = 2^(-v^2 log2(e) / 2) * 0.3989...
  return fmaf(v * 0.39894228040143267794f, __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f), cdf);
}

// Blocks b and b + 8 share an XCD (round-robin dispatch).  Remap so that consecutive logical
// tiles — which share the same activation columns — land on one XCD's L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = id & 7, local = id >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}
#endif

// hipFuncSetAttribute(..., MaxDynamicSharedMemorySize) takes effect on the current device only: a call
// site keeps one of these (static) and the attribute is set once per device that launches from it.
struct DynLdsOnce {
  std::atomic<uint64_t> devices{0};   // bit d: done on device d (devices past 63 set it every time)
};
void allow_dynamic_lds(DynLdsOnce& once, const void* kernel, int bytes);

struct Scope {
  Scope(const char* name, double flops, double bytes, hipStream_t s);
  ~Scope();
  void* h;
};

}  // namespace jirf
