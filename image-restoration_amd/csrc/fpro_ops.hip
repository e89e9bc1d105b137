// FPro's frequency-prompt operators for gfx950 (reference basicsr/models/archs/fpro_arch.py):
//   dynamic_filter_channel        :339-375   per-image, per-group 3x3 low-pass with reflection padding
//   frequenctSpecificPromptGenetator :378-449  high branch (1x3 / 3x1 depthwise + GELU + products);
//                                   the low branch reaches the network only through its spatial MEAN
//                                   (see prompt_low_kernel), so no FFT is executed
//   highFrequencyPromptFusion     :288-336   8x8-window cross attention with the literal
//                                   transpose-then-reshape flatten of :331-332
//   lowFrequencyPromptFusion      :212-246   one pooled key / value token: softmax over one key == 1
//   nn.interpolate bilinear / bicubic (align_corners=False), torch-equivalent index arithmetic
#include "common.h"

namespace jirf {

__device__ __forceinline__ float gelu_erf_f(float v) { return gelu_fast(v); }

// ---- per-channel spatial mean: out[b][c] = mean_p x[b][c][p] -------------------------------------
__global__ __launch_bounds__(256) void channel_mean_kernel(const float* __restrict__ x, long x_bs,
                                                           float* __restrict__ out, int C, long P) {
  __shared__ float red[256];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* __restrict__ xp = x + (long)b * x_bs + (long)c * P;
  float s = 0.f;
  for (long i = tid; i < P; i += 256) s += xp[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[(long)b * C + c] = red[0] / (float)P;
}

void launch_channel_mean(TV x, float* out, int B, int C, long P, hipStream_t s) {
  hipLaunchKernelGGL(channel_mean_kernel, dim3(C, B), dim3(256), 0, s, x.p, x.bs, out, C, P);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- dynamic filter coefficients: pooled[C] -> 1x1 -> sigmoid gate -> BatchNorm (eval) -> softmax
//      over the 9 taps of each group.  One workgroup per image; coeff [B][G*9].
__global__ __launch_bounds__(128) void dynfilter_coeff_kernel(const float* __restrict__ pooled,
                                                              const float* __restrict__ w_conv,   // [G9][C]
                                                              const float* __restrict__ w_gate,   // [G9][G9]
                                                              const float* __restrict__ bn_w,
                                                              const float* __restrict__ bn_b,
                                                              const float* __restrict__ bn_mean,
                                                              const float* __restrict__ bn_var,
                                                              float* __restrict__ coeff, int C, int G) {
  __shared__ float f[128], g2[128];
  const int b = blockIdx.x, t = threadIdx.x, n = G * 9;
  if (t < n) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(w_conv[t * C + c], pooled[(long)b * C + c], a);
    f[t] = a;
  }
  __syncthreads();
  if (t < n) {
    float a = 0.f;
    for (int j = 0; j < n; ++j) a = fmaf(w_gate[t * n + j], f[j], a);
    float v = f[t] * (1.0f / (1.0f + expf(-a)));
    v = (v - bn_mean[t]) / sqrtf(bn_var[t] + 1e-5f) * bn_w[t] + bn_b[t];
    g2[t] = v;
  }
  __syncthreads();
  if (t < n) {
    const int g0 = (t / 9) * 9;
    float m = -INFINITY;
    for (int j = 0; j < 9; ++j) m = fmaxf(m, g2[g0 + j]);
    float s = 0.f;
    for (int j = 0; j < 9; ++j) s += expf(g2[g0 + j] - m);
    coeff[(long)b * n + t] = expf(g2[t] - m) / s;
  }
}

// low[b][c][y][x] = sum_tap coeff[b][group(c)][tap] * x[b][c][reflect(y+dy)][reflect(x+dx)]; high = x - low
__global__ __launch_bounds__(256) void dynfilter_apply_kernel(const float* __restrict__ x, long x_bs,
                                                              const float* __restrict__ coeff,
                                                              float* __restrict__ low, float* __restrict__ high,
                                                              long o_bs, int C, int G, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  const long P = (long)H * W;
  if (i >= P) return;
  const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
  const float* __restrict__ k = coeff + ((long)b * G + c / (C / G)) * 9;
  const float* __restrict__ plane = x + (long)b * x_bs + (long)c * P;
  float acc = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    int ry = yy + dy;
    ry = ry < 0 ? -ry : (ry >= H ? 2 * (H - 1) - ry : ry);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      int rx = xx + dx;
      rx = rx < 0 ? -rx : (rx >= W ? 2 * (W - 1) - rx : rx);
      acc = fmaf(k[(dy + 1) * 3 + dx + 1], plane[(long)ry * W + rx], acc);
    }
  }
  const long o = (long)b * o_bs + (long)c * P + i;
  low[o] = acc;
  high[o] = plane[i] - acc;
}

void launch_dynfilter(TV x, const DynFilterW& w, float* pooled, float* coeff, TV low, TV high, int B, int C,
                      int H, int W, hipStream_t s) {
  JIRF_REQUIRE(C % w.groups == 0 && w.groups * 9 <= 128 && H >= 2 && W >= 2, JIRF_EINVAL,
               "dynamic filter: C=%d groups=%d on %dx%d", C, w.groups, H, W);
  JIRF_REQUIRE(low.bs == high.bs, JIRF_EINVAL, "dynamic filter: low / high strides differ");
  const long P = (long)H * W;
  Scope sc(intern_label("dynfilter C%d H%d W%d", C, H, W), 20.0 * B * C * (double)P, 16.0 * B * C * (double)P, s);
  launch_channel_mean(x, pooled, B, C, P, s);
  hipLaunchKernelGGL(dynfilter_coeff_kernel, dim3(B), dim3(128), 0, s, pooled, w.conv, w.gate, w.bn_w, w.bn_b,
                     w.bn_mean, w.bn_var, coeff, C, w.groups);
  hipLaunchKernelGGL(dynfilter_apply_kernel, dim3((unsigned)((P + 255) / 256), C, B), dim3(256), 0, s, x.p, x.bs,
                     coeff, low.p, high.p, low.bs, C, w.groups, H, W);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- resize, align_corners = False ---------------------------------------------------------------
// bilinear: src = max(0, (dst + 0.5) * in / out - 0.5), two taps.  bicubic: src unclamped, four taps
// with clamped indices, cubic convolution coefficients with A = -0.75.
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <bool CUBIC>
__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ x, long x_bs, long x_cs,
                                                     float* __restrict__ y, long y_bs, int C, int Hi, int Wi,
                                                     int Ho, int Wo, float sh, float sw) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (i >= (long)Ho * Wo) return;
  const int oy = (int)(i / Wo), ox = (int)(i - (long)oy * Wo);
  const float* __restrict__ plane = x + (long)b * x_bs + (long)c * x_cs;
  float out;
  if constexpr (!CUBIC) {
    const float fy = fmaxf(((float)oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)ox + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    out = hy * (hx * plane[(long)y0 * Wi + x0] + lx * plane[(long)y0 * Wi + x1]) +
          ly * (hx * plane[(long)y1 * Wi + x0] + lx * plane[(long)y1 * Wi + x1]);
  } else {
    const float fy = ((float)oy + 0.5f) * sh - 0.5f, fx = ((float)ox + 0.5f) * sw - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int iy = (int)fly, ix = (int)flx;
    float wy[4], wx[4];
    cubic_coeffs(fy - fly, wy);
    cubic_coeffs(fx - flx, wx);
    out = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), Hi - 1);
      float r = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int xx = min(max(ix - 1 + d, 0), Wi - 1);
        r = fmaf(wx[d], plane[(long)yy * Wi + xx], r);
      }
      out = fmaf(wy[a], r, out);
    }
  }
  y[(long)b * y_bs + ((long)c * Ho + oy) * Wo + ox] = out;
}

void launch_resize(const float* x, long x_bs, long x_cs, float* y, long y_bs, int B, int C, int Hi, int Wi, int Ho,
                   int Wo, bool cubic, hipStream_t s) {
  JIRF_REQUIRE(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, JIRF_EINVAL, "resize %dx%d -> %dx%d", Hi, Wi, Ho, Wo);
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const dim3 grid((unsigned)(((long)Ho * Wo + 255) / 256), C, B), block(256);
  Scope sc(cubic ? "resize_bicubic" : "resize_bilinear", 0.0, 4.0 * B * C * ((double)Hi * Wi + (double)Ho * Wo), s);
  if (cubic)
    hipLaunchKernelGGL(resize_kernel<true>, grid, block, 0, s, x, x_bs, x_cs, y, y_bs, C, Hi, Wi, Ho, Wo, sh, sw);
  else
    hipLaunchKernelGGL(resize_kernel<false>, grid, block, 0, s, x, x_bs, x_cs, y, y_bs, C, Hi, Wi, Ho, Wo, sh, sw);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- high-frequency prompt: y = gelu(dw3x1(dw1x3(f) + b1) + b2) * f * wmap -------------------------
// Both depthwise convs pad with zeros: a row of the intermediate that lies outside the image
// contributes 0, not b1.
__global__ __launch_bounds__(256) void fspg_high_kernel(const float* __restrict__ f, long f_bs,
                                                        const float* __restrict__ k1, const float* __restrict__ b1,
                                                        const float* __restrict__ k2, const float* __restrict__ b2,
                                                        const float* __restrict__ wmap, float* __restrict__ y,
                                                        long y_bs, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  const long P = (long)H * W;
  if (i >= P) return;
  const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
  const float* __restrict__ plane = f + (long)b * f_bs + (long)c * P;
  const float a0 = k1[c * 3], a1 = k1[c * 3 + 1], a2 = k1[c * 3 + 2], bb1 = b1[c];
  float t = b2[c];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int ry = yy + dy;
    if (ry < 0 || ry >= H) continue;
    const float* __restrict__ row = plane + (long)ry * W;
    float u = bb1 + a1 * row[xx];
    if (xx > 0) u = fmaf(a0, row[xx - 1], u);
    if (xx + 1 < W) u = fmaf(a2, row[xx + 1], u);
    t = fmaf(k2[c * 3 + dy + 1], u, t);
  }
  y[(long)b * y_bs + (long)c * P + i] = gelu_erf_f(t) * plane[i] * wmap[(long)c * P + i];
}

void launch_fspg_high(TV f, const float* k1, const float* b1, const float* k2, const float* b2, const float* wmap,
                      TV y, int B, int C, int H, int W, hipStream_t s) {
  const long P = (long)H * W;
  Scope sc(intern_label("fspg_high C%d H%d W%d", C, H, W), 30.0 * B * C * (double)P, 12.0 * B * C * (double)P, s);
  hipLaunchKernelGGL(fspg_high_kernel, dim3((unsigned)((P + 255) / 256), C, B), dim3(256), 0, s, f.p, f.bs, k1, b1, k2,
                     b2, wmap, y.p, y.bs, H, W);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- 8x8-window cross attention --------------------------------------------------------------------
// One wave per (window, head): lane n is query token n.  q [C][P], kv [2C][P] planar (k = first C
// channels, v = last C; channel = head * hd + d).  K then V are staged through the wave's LDS slice.
// The product out[head][n][d] is written where the reference's transpose + reshape puts it: flat
// index f = (head * hd + d) * 64 + n  ->  token f / C, channel f % C of the same window.
constexpr int CW_MAXD = 96;

template <int HD>
__global__ __launch_bounds__(256) void cross_win_attn_kernel(const float* __restrict__ q, long q_bs,
                                                             const float* __restrict__ kv, long kv_bs,
                                                             float* __restrict__ out, long o_bs, int C, int heads,
                                                             int H, int W, float scale, int n_tasks) {
  constexpr int LD = HD + 4;
  extern __shared__ __attribute__((aligned(16))) float cw_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // neighbouring windows share an XCD's L2
  if (task >= n_tasks) return;   // whole wave; only wave-level synchronisation below
  float* __restrict__ T = cw_smem + wave * 64 * LD;
  const int nwx = W >> 3, nW = (H >> 3) * nwx;
  const long P = (long)H * W;
  const int gw = task / heads, h = task - gw * heads;
  const int img = gw / nW, w = gw - img * nW;
  const int wy = w / nwx, wx = w - wy * nwx;
  auto pixel = [&](int n) { return (long)(wy * 8 + (n >> 3)) * W + wx * 8 + (n & 7); };
  const long pq = pixel(lane);
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  float qr[HD];
  {
    const float* __restrict__ qp = q + (long)img * q_bs + (long)(h * HD) * P + pq;
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = qp[(long)d * P] * scale;
    const float* __restrict__ kp = kv + (long)img * kv_bs + (long)(h * HD) * P + pq;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) T[lane * LD + d] = kp[(long)d * P];
  }
  wave_sync();
  float sc[64];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < 64; ++m) {
    const float4* __restrict__ kr = reinterpret_cast<const float4*>(&T[m * LD]);
    float s = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const float4 k4 = kr[d4];
      s = fmaf(qr[4 * d4 + 0], k4.x, s);
      s = fmaf(qr[4 * d4 + 1], k4.y, s);
      s = fmaf(qr[4 * d4 + 2], k4.z, s);
      s = fmaf(qr[4 * d4 + 3], k4.w, s);
    }
    sc[m] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < 64; ++m) {
    sc[m] = fast_exp(sc[m] - mx);
    sum += sc[m];
  }
  const float inv = 1.0f / sum;
  wave_sync();   // every lane is done reading K
  {
    const float* __restrict__ vp = kv + (long)img * kv_bs + (long)(C + h * HD) * P + pq;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) T[lane * LD + d] = vp[(long)d * P];
  }
  wave_sync();
  float* __restrict__ ob = out + (long)img * o_bs;
  if constexpr (HD <= 48) if (64 % heads == 0) {
    // All HD accumulators fit in registers: the wave's outputs are exchanged through its LDS tile so that
    // the stores run along pixels.  The flat index f = (h * HD + d) * 64 + n of this wave covers the
    // 64 / heads tokens [h * 64 / heads, ...) with all C channels; written per lane they would go to 64
    // different channel planes (one cache line each).
    float o[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) o[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const float p = sc[m] * inv;
      const float4* __restrict__ vr = reinterpret_cast<const float4*>(&T[m * LD]);
#pragma unroll
      for (int e4 = 0; e4 < HD / 4; ++e4) {
        const float4 v4 = vr[e4];
        o[4 * e4 + 0] = fmaf(p, v4.x, o[4 * e4 + 0]);
        o[4 * e4 + 1] = fmaf(p, v4.y, o[4 * e4 + 1]);
        o[4 * e4 + 2] = fmaf(p, v4.z, o[4 * e4 + 2]);
        o[4 * e4 + 3] = fmaf(p, v4.w, o[4 * e4 + 3]);
      }
    }
    wave_sync();   // every lane is done reading V
#pragma unroll
    for (int e = 0; e < HD; ++e) T[e * 64 + lane] = o[e];   // local flat index f' = d * 64 + n
    wave_sync();
    const int tpw = 64 / heads;            // tokens this wave produces (heads divides 64)
    const int tok0 = h * tpw;
    for (int idx = lane; idx < 64 * HD; idx += 64) {
      const int c2 = idx / tpw, tl = idx - c2 * tpw;
      ob[(long)c2 * P + pixel(tok0 + tl)] = T[tl * C + c2];
    }
    return;
  }
  // the output in chunks of 16 dims keeps the accumulators in registers for any HD
#pragma unroll
  for (int d0 = 0; d0 < HD; d0 += 16) {
    float o[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const float p = sc[m] * inv;
      const float4* __restrict__ vr = reinterpret_cast<const float4*>(&T[m * LD + d0]);
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const float4 v4 = vr[e4];
        o[4 * e4 + 0] = fmaf(p, v4.x, o[4 * e4 + 0]);
        o[4 * e4 + 1] = fmaf(p, v4.y, o[4 * e4 + 1]);
        o[4 * e4 + 2] = fmaf(p, v4.z, o[4 * e4 + 2]);
        o[4 * e4 + 3] = fmaf(p, v4.w, o[4 * e4 + 3]);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = (h * HD + d0 + e) * 64 + lane;
      const int n2 = f / C, c2 = f - n2 * C;
      ob[(long)c2 * P + pixel(n2)] = o[e];
    }
  }
}

// ---- matrix-core form of the cross attention (head dims up to 64, heads dividing 64) ---------------------
// The vector-ALU kernel above reads every K / V row as 16-byte LDS broadcasts and is bound by LDS
// bandwidth.  Here S^T[m][n] = sum_d K[m][d] Q[n][d] runs on fp32 MFMA (32x32x2) with both operands
// loaded from global in fragment layout; a query's softmax then lives in one lane's column (32 registers
// plus the partner lane).  O^T[d][n] = sum_m V[m][d] P[m][n] takes the probability registers as its B
// operand in place and V[m][d] from the wave's LDS tile (zero columns pad the head dim to 32 d-rows per
// tile).  The outputs leave through the same LDS tile in the reference's flat order, as above.
typedef float cw_f32x16 __attribute__((ext_vector_type(16)));
constexpr int CWM_LD = 72;   // V tile row stride: rows m and m + 4 on disjoint bank halves; >= 64 columns

template <int HD>
__global__ __launch_bounds__(256, 2) void cross_win_attn_mfma_kernel(const float* __restrict__ q, long q_bs,
                                                                     const float* __restrict__ kv, long kv_bs,
                                                                     float* __restrict__ out, long o_bs, int C,
                                                                     int heads, int H, int W, float scale,
                                                                     int n_tasks) {
  constexpr int ET = (HD + 31) / 32;   // 32-row tiles of the output's d axis
  constexpr int KS = HD / 2;           // MFMA steps of the score product
  __shared__ float Tile[4][64 * CWM_LD];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (task >= n_tasks) return;   // whole wave; only wave-level synchronisation below
  float* __restrict__ T = Tile[wave];
  const int nwx = W >> 3, nW = (H >> 3) * nwx;
  const long P = (long)H * W;
  const int gw = task / heads, h = task - gw * heads;
  const int img = gw / nW, w = gw - img * nW;
  const int wy = w / nwx, wx = w - wy * nwx;
  auto pixel = [&](int n) { return (long)(wy * 8 + (n >> 3)) * W + wx * 8 + (n & 7); };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  float kf[2][KS], qf[2][KS];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const long px = pixel(t2 * 32 + l31);
    const float* __restrict__ kp = kv + (long)img * kv_bs + (long)(h * HD + half) * P + px;
    const float* __restrict__ qp = q + (long)img * q_bs + (long)(h * HD + half) * P + px;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[t2][s] = kp[(long)(2 * s) * P];
      qf[t2][s] = qp[(long)(2 * s) * P] * scale;
    }
  }
  {
    const float* __restrict__ vp = kv + (long)img * kv_bs + (long)(C + h * HD) * P + pixel(lane);
#pragma unroll 8
    for (int d = 0; d < HD; ++d) T[lane * CWM_LD + d] = vp[(long)d * P];
#pragma unroll
    for (int d = HD; d < ET * 32; ++d) T[lane * CWM_LD + d] = 0.f;
  }
  cw_f32x16 S[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[mt][nt][r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        S[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[mt][s], qf[nt][s], S[mt][nt], 0, 0, 0);
  // softmax over the keys, per query column
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[mt][nt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[mt][nt][r] = fast_exp(S[mt][nt][r] - mx);
        sum += S[mt][nt][r];
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[mt][nt][r] *= inv;
  }
  wave_sync();   // the V tile is complete
  cw_f32x16 O[ET][2];
#pragma unroll
  for (int et = 0; et < ET; ++et)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[et][nt][r] = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        const float av = T[m * CWM_LD + et * 32 + l31];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          O[et][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, S[mt][nt][r], O[et][nt], 0, 0, 0);
      }
    }
  wave_sync();   // every lane is done reading V: the tile now carries the outputs, flat index d * 64 + n
#pragma unroll
  for (int et = 0; et < ET; ++et)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = et * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (d < HD) T[d * 64 + nt * 32 + l31] = O[et][nt][r];
      }
  wave_sync();
  float* __restrict__ ob = out + (long)img * o_bs;
  const int tpw = 64 / heads, tok0 = h * tpw;
  for (int idx = lane; idx < 64 * HD; idx += 64) {
    const int c2 = idx / tpw, tl = idx - c2 * tpw;
    ob[(long)c2 * P + pixel(tok0 + tl)] = T[tl * C + c2];
  }
}

void launch_cross_window_attention(TV q, TV kv, TV out, int B, int C, int heads, int H, int W, hipStream_t s) {
  JIRF_REQUIRE(heads > 0 && C % heads == 0, JIRF_EINVAL, "cross window attention: C=%d heads=%d", C, heads);
  const int hd = C / heads;
  JIRF_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8, JIRF_EINVAL,
               "cross window attention: %dx%d is not a multiple of the 8x8 window", H, W);
  const int n_tasks = B * (H / 8) * (W / 8) * heads;
  const float scale = 1.0f / sqrtf((float)hd);
  const double px = (double)B * H * W;
  Scope sc(intern_label("cross_win_attn C%d H%d W%d", C, H, W), 4.0 * 64 * C * px, 16.0 * C * px, s);
  const dim3 grid((unsigned)((n_tasks + 3) / 4)), block(256);
#define JIRF_CW_CASE(HD)                                                                                  \
  do {                                                                                                    \
    const size_t smem = (size_t)4 * 64 * (HD + 4) * sizeof(float);                                        \
    static DynLdsOnce once;                                        \
    allow_dynamic_lds(once, reinterpret_cast<const void*>(cross_win_attn_kernel<HD>), 128 * 1024); \
    hipLaunchKernelGGL((cross_win_attn_kernel<HD>), grid, block, smem, s, q.p, q.bs, kv.p, kv.bs, out.p,  \
                       out.bs, C, heads, H, W, scale, n_tasks);                                           \
  } while (0)
  static const bool valu = std::getenv("JIRF_CROSSWIN_VALU") != nullptr;
  if (!valu && 64 % heads == 0 && (hd == 16 || hd == 32 || hd == 48 || hd == 64)) {
#define JIRF_CWM_CASE(HD)                                                                                 \
  hipLaunchKernelGGL((cross_win_attn_mfma_kernel<HD>), grid, block, 0, s, q.p, q.bs, kv.p, kv.bs, out.p,  \
                     out.bs, C, heads, H, W, scale, n_tasks)
    if (hd == 16) JIRF_CWM_CASE(16);
    else if (hd == 32) JIRF_CWM_CASE(32);
    else if (hd == 48) JIRF_CWM_CASE(48);
    else JIRF_CWM_CASE(64);
#undef JIRF_CWM_CASE
    JIRF_HIP_CHECK(hipGetLastError());
    return;
  }
  switch (hd) {
    case 16: JIRF_CW_CASE(16); break;
    case 32: JIRF_CW_CASE(32); break;
    case 48: JIRF_CW_CASE(48); break;
    case 64: JIRF_CW_CASE(64); break;
    case 96: JIRF_CW_CASE(96); break;
    default:
      JIRF_REQUIRE(false, JIRF_EINVAL, "cross window attention: head dim %d is not built (16, 32, 48, 64, 96)", hd);
  }
#undef JIRF_CW_CASE
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- low-frequency prompt, reduced to what reaches the network ------------------------------------
// lowFrequencyPromptFusion pools its prompt to ONE token, so its softmax runs over a single key and
// its output is project_out(value) broadcast over the pixels; the value depends on the prompt only
// through its spatial mean, and the mean of ifft2(Y).real is Re(Y[0,0]) / (H W).  Y[0,0] before the
// gate is (sum over pixels of the resized low part, 0).  So per image:
//   s[c]     = sum_p resized_low[c][p]                         (given as mean[c] * H W)
//   a        = gelu(Wb [s; 0] + bb)                            (the 1x1 conv + GELU gate, 2C2 -> 2C2)
//   re[c]    = s[c] a[c],  im[c] = 0 * a[C2 + c] = 0
//   m[c]     = (re[c] wr[c] - im[c] wi[c]) * ifft_scale        (complex weight sample at (0,0))
//   bias[o]  = sum_c M[o][c] m[c] + b0[o]                      (M = Wfusion_lo Wproj_lo Wkv_value, folded on
//                                                               the host; b0 = fusion bias + folded terms)
// bias is the per-image vector the fusion 1x1 conv adds to every pixel.
__global__ __launch_bounds__(256) void prompt_low_kernel(const float* __restrict__ mean, float hw,
                                                         const float* __restrict__ Wb, const float* __restrict__ bb,
                                                         const float* __restrict__ w00, float ifft_scale,
                                                         const float* __restrict__ M, const float* __restrict__ b0,
                                                         float* __restrict__ bias_out, int C2, int C) {
  __shared__ float sv[128], mv[128];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < C2) sv[t] = mean[(long)b * C2 + t] * hw;
  __syncthreads();
  if (t < C2) {
    float a = bb[t];
    for (int j = 0; j < C2; ++j) a = fmaf(Wb[(long)t * 2 * C2 + j], sv[j], a);   // imaginary inputs are 0
    const float re = sv[t] * gelu_erf_f(a);
    mv[t] = re * w00[t] * ifft_scale;                                            // im = 0: - im * wi drops out
  }
  __syncthreads();
  for (int o = t; o < C; o += 256) {
    float a = b0[o];
    for (int c = 0; c < C2; ++c) a = fmaf(M[(long)o * C2 + c], mv[c], a);
    bias_out[(long)b * C + o] = a;
  }
}

void launch_prompt_low(const float* mean, float hw, const float* Wb, const float* bb, const float* w00,
                       float ifft_scale, const float* M, const float* b0, float* bias_out, int B, int C2, int C,
                       hipStream_t s) {
  JIRF_REQUIRE(C2 <= 128, JIRF_EINVAL, "prompt low branch: %d prompt channels (at most 128)", C2);
  hipLaunchKernelGGL(prompt_low_kernel, dim3(B), dim3(256), 0, s, mean, hw, Wb, bb, w00, ifft_scale, M, b0, bias_out,
                     C2, C);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace jirf
