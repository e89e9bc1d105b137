// HBM-bound kernels of libjirf_hip (gfx950): channel LayerNorm, its statistics-only form, and the
// depthwise 3x3 convolution with the GDFN gate fused in.
//
// Activations are planar [channel][pixel]; a wave's 64 lanes always sit on 64 (or 256) adjacent
// pixels, so every global access is a full 256-B / 1-KiB row segment.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace jirf {

__device__ __forceinline__ float gelu_erf_pw(float v) { return gelu_fast(v); }

// ---- per-pixel LayerNorm over channels ---------------------------------------------------------
// One lane per pixel.  Pass 1 sums the channels, pass 2 sums squared deviations (the two-pass
// population variance the reference's `var(unbiased=False)` denotes), pass 3 writes; passes 2/3
// re-read lines that pass 1 left in L2.
// kind: 1 = WithBias ((x - mu) * rstd * w + b), 2 = BiasFree (x * rstd * w), 0 = statistics only.
template <int KIND>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, long y_bs,
                                                        float* __restrict__ mu_out,
                                                        float* __restrict__ rstd_out, int C, int P) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= P) return;
  const float* __restrict__ xp = x + (long)b * x_bs + p;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += xp[(long)c * P];
  const float mu = s / (float)C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) {
    const float d = xp[(long)c * P] - mu;
    v = fmaf(d, d, v);
  }
  const float rstd = 1.0f / sqrtf(v / (float)C + 1e-5f);
  if constexpr (KIND == 0) {
    mu_out[(long)b * P + p] = mu;
    rstd_out[(long)b * P + p] = rstd;
  } else {
    float* __restrict__ yp = y + (long)b * y_bs + p;
    for (int c = 0; c < C; ++c) {
      const float xv = xp[(long)c * P];
      if constexpr (KIND == 1)
        yp[(long)c * P] = (xv - mu) * rstd * w[c] + bias[c];
      else
        yp[(long)c * P] = xv * rstd * w[c];
    }
  }
}

// Stand-alone LayerNorm, channels dealt over the 4 waves of a workgroup: a workgroup owns 64 * PX
// adjacent pixels (PX = 4: one float4 per lane), every wave reduces its channel slice, the partial
// sums meet in LDS, and each wave writes its own slice.  Passes 2 and 3 re-read lines pass 1 left
// in L2.  Same two-pass statistics as layernorm_kernel.
template <int KIND, int PX>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, long x_bs,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ y, long y_bs, int C, int P) {
  __shared__ float part[2][4][64 * PX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = (blockIdx.x * 64 + lane) * PX;
  const int b = blockIdx.y;
  const bool ok = p < P;   // PX == 4 requires P % 4 == 0, so a lane is all in or all out
  const float* __restrict__ xp = x + (long)b * x_bs + (ok ? p : 0);
  float* __restrict__ yp = y + (long)b * y_bs + (ok ? p : 0);
  float v[PX], s[PX];
  auto load = [&](int c) {
    if constexpr (PX == 4) {
      const float4 t = *reinterpret_cast<const float4*>(xp + (long)c * P);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      v[0] = xp[(long)c * P];
    }
  };
#pragma unroll
  for (int i = 0; i < PX; ++i) s[i] = 0.f;
#pragma unroll 4
  for (int c = wave; c < C; c += 4) {
    load(c);
#pragma unroll
    for (int i = 0; i < PX; ++i) s[i] += v[i];
  }
#pragma unroll
  for (int i = 0; i < PX; ++i) part[0][wave][lane * PX + i] = s[i];
  __syncthreads();
  float mu[PX], rstd[PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int q = lane * PX + i;
    mu[i] = (part[0][0][q] + part[0][1][q] + part[0][2][q] + part[0][3][q]) / (float)C;
    s[i] = 0.f;
  }
#pragma unroll 4
  for (int c = wave; c < C; c += 4) {
    load(c);
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      const float d = v[i] - mu[i];
      s[i] = fmaf(d, d, s[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < PX; ++i) part[1][wave][lane * PX + i] = s[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int q = lane * PX + i;
    rstd[i] = 1.0f / sqrtf((part[1][0][q] + part[1][1][q] + part[1][2][q] + part[1][3][q]) / (float)C + 1e-5f);
  }
  if (!ok) return;
#pragma unroll 4
  for (int c = wave; c < C; c += 4) {
    load(c);
    const float g = w[c];
    float o[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      if constexpr (KIND == 1) o[i] = (v[i] - mu[i]) * rstd[i] * g + bias[c];
      else o[i] = v[i] * rstd[i] * g;
    }
    if constexpr (PX == 4) *reinterpret_cast<float4*>(yp + (long)c * P) = make_float4(o[0], o[1], o[2], o[3]);
    else yp[(long)c * P] = o[0];
  }
}

// Single-pass form for C <= 4 * NCH: each wave keeps its channel slice (channels wave, wave + 4, ...) of
// the 256 pixels in registers, so x is read from memory once, with all NCH 16-byte loads of a lane in flight
// together; the two-pass statistics (mean, then variance about it) run on the registers.
template <int KIND, int NCH>
__global__ __launch_bounds__(256) void layernorm_regs_kernel(const float* __restrict__ x, long x_bs,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, long y_bs, int C, int P) {
  __shared__ float part[2][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = (blockIdx.x * 64 + lane) * 4;
  const int b = blockIdx.y;
  const bool ok = p < P;   // P % 4 == 0: a lane is all in or all out
  const float* __restrict__ xp = x + (long)b * x_bs + (ok ? p : 0);
  float4 v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = wave + 4 * i;
    v[i] = *reinterpret_cast<const float4*>(xp + (long)min(c, C - 1) * P);   // clamped: no predicate around the load
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (wave + 4 * i < C) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
  *reinterpret_cast<float4*>(&part[0][wave][lane * 4]) = s;
  __syncthreads();
  float4 mu;
  {
    const float4 a0 = *reinterpret_cast<const float4*>(&part[0][0][lane * 4]), a1 = *reinterpret_cast<const float4*>(&part[0][1][lane * 4]);
    const float4 a2 = *reinterpret_cast<const float4*>(&part[0][2][lane * 4]), a3 = *reinterpret_cast<const float4*>(&part[0][3][lane * 4]);
    mu = make_float4((a0.x + a1.x + a2.x + a3.x) / (float)C, (a0.y + a1.y + a2.y + a3.y) / (float)C,
                     (a0.z + a1.z + a2.z + a3.z) / (float)C, (a0.w + a1.w + a2.w + a3.w) / (float)C);
  }
  s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (wave + 4 * i < C) {
      const float dx = v[i].x - mu.x, dy = v[i].y - mu.y, dz = v[i].z - mu.z, dw = v[i].w - mu.w;
      s.x = fmaf(dx, dx, s.x); s.y = fmaf(dy, dy, s.y); s.z = fmaf(dz, dz, s.z); s.w = fmaf(dw, dw, s.w);
    }
  *reinterpret_cast<float4*>(&part[1][wave][lane * 4]) = s;
  __syncthreads();
  float4 rs;
  {
    const float4 a0 = *reinterpret_cast<const float4*>(&part[1][0][lane * 4]), a1 = *reinterpret_cast<const float4*>(&part[1][1][lane * 4]);
    const float4 a2 = *reinterpret_cast<const float4*>(&part[1][2][lane * 4]), a3 = *reinterpret_cast<const float4*>(&part[1][3][lane * 4]);
    rs = make_float4(1.0f / sqrtf((a0.x + a1.x + a2.x + a3.x) / (float)C + 1e-5f), 1.0f / sqrtf((a0.y + a1.y + a2.y + a3.y) / (float)C + 1e-5f),
                     1.0f / sqrtf((a0.z + a1.z + a2.z + a3.z) / (float)C + 1e-5f), 1.0f / sqrtf((a0.w + a1.w + a2.w + a3.w) / (float)C + 1e-5f));
  }
  if (!ok) return;
  float* __restrict__ yp = y + (long)b * y_bs + p;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = wave + 4 * i;
    if (c >= C) break;
    const float g = w[c];
    float4 o;
    if constexpr (KIND == 1) {
      const float be = bias[c];
      o = make_float4((v[i].x - mu.x) * rs.x * g + be, (v[i].y - mu.y) * rs.y * g + be, (v[i].z - mu.z) * rs.z * g + be,
                      (v[i].w - mu.w) * rs.w * g + be);
    } else {
      o = make_float4(v[i].x * rs.x * g, v[i].y * rs.y * g, v[i].z * rs.z * g, v[i].w * rs.w * g);
    }
    *reinterpret_cast<float4*>(yp + (long)c * P) = o;
  }
}

// Small planes with many channels (the deep levels of a single image): the forms above walk a wave's channel slice
// one load after another, C / 4 deep and three times over, on a handful of workgroups.  Here a workgroup owns 16
// adjacent pixels and deals the channels over 16 thread rows, so a thread holds C / 16 values (all of its loads in
// flight together, x read once) and the statistics meet in LDS.  Two-pass mean / variance as everywhere else.
template <int KIND, int NCH>
__global__ __launch_bounds__(256) void layernorm_small_kernel(const float* __restrict__ x, long x_bs,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ y, long y_bs, int C, int P,
                                                              float* __restrict__ mu_out, float* __restrict__ rstd_out) {
  __shared__ float part[2][16][17];
  const int px = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int p = blockIdx.x * 16 + px, b = blockIdx.y;
  const bool ok = p < P;
  const float* __restrict__ xp = x + (long)b * x_bs + (ok ? p : 0);
  float v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) v[i] = xp[(long)min(cl + 16 * i, C - 1) * P];   // clamped: no predicate around the load
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (cl + 16 * i < C) sum += v[i];
  part[0][cl][px] = sum;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) mu += part[0][j][px];
  mu /= (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (cl + 16 * i < C) sq = fmaf(v[i] - mu, v[i] - mu, sq);
  part[1][cl][px] = sq;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) var += part[1][j][px];
  const float rstd = 1.0f / sqrtf(var / (float)C + 1e-5f);
  if (!ok) return;
  if constexpr (KIND == 0) {   // statistics only
    if (cl == 0) {
      mu_out[(long)b * P + p] = mu;
      rstd_out[(long)b * P + p] = rstd;
    }
    return;
  }
  float* __restrict__ yp = y + (long)b * y_bs + p;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = cl + 16 * i;
    if (c >= C) break;
    if constexpr (KIND == 1) yp[(long)c * P] = (v[i] - mu) * rstd * w[c] + bias[c];
    else yp[(long)c * P] = v[i] * rstd * w[c];
  }
}

void launch_layernorm(TV x, const float* w, const float* b, TV y, int B, int C, int P, int kind,
                      hipStream_t s) {
  JIRF_REQUIRE(kind == 1 || kind == 2, JIRF_EINVAL, "layernorm: kind %d", kind);
  JIRF_REQUIRE(kind == 2 || b != nullptr, JIRF_EINVAL, "layernorm: WithBias needs a bias");
  Scope sc(intern_label("layernorm C%d P%d", C, P), 8.0 * B * (double)C * P, 8.0 * B * (double)C * P, s);
  const bool vec = P % 4 == 0 && x.bs % 4 == 0 && y.bs % 4 == 0 && (reinterpret_cast<uintptr_t>(x.p) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(y.p) & 15) == 0 && (long)B * P >= 256L * 1024;
  const dim3 block(256);
  if ((long)B * P <= 16384 && C >= 128 && C <= 1024) {
    const dim3 grid((P + 15) / 16, B);
#define JIRF_LN_SMALL(N)                                                                                              \
  do {                                                                                                                \
    if (kind == 1) hipLaunchKernelGGL((layernorm_small_kernel<1, N>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P, nullptr, nullptr); \
    else hipLaunchKernelGGL((layernorm_small_kernel<2, N>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P, nullptr, nullptr);       \
  } while (0)
    if (C <= 256) JIRF_LN_SMALL(16);
    else if (C <= 512) JIRF_LN_SMALL(32);
    else JIRF_LN_SMALL(64);
#undef JIRF_LN_SMALL
  } else if (vec && C <= 96) {
    const dim3 grid((P + 255) / 256, B);
#define JIRF_LN_REGS(N)                                                                                              \
  do {                                                                                                               \
    if (kind == 1) hipLaunchKernelGGL((layernorm_regs_kernel<1, N>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P); \
    else hipLaunchKernelGGL((layernorm_regs_kernel<2, N>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P);       \
  } while (0)
    if (C <= 32) JIRF_LN_REGS(8);
    else if (C <= 64) JIRF_LN_REGS(16);
    else JIRF_LN_REGS(24);
#undef JIRF_LN_REGS
  } else if (vec) {
    const dim3 grid((P + 255) / 256, B);
    if (kind == 1)
      hipLaunchKernelGGL((layernorm_split_kernel<1, 4>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P);
    else
      hipLaunchKernelGGL((layernorm_split_kernel<2, 4>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P);
  } else {
    const dim3 grid((P + 63) / 64, B);
    if (kind == 1)
      hipLaunchKernelGGL((layernorm_split_kernel<1, 1>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P);
    else
      hipLaunchKernelGGL((layernorm_split_kernel<2, 1>), grid, block, 0, s, x.p, x.bs, w, b, y.p, y.bs, C, P);
  }
  JIRF_HIP_CHECK(hipGetLastError());
}

// Statistics only, for small planes: 64 pixels per workgroup, the channels dealt over its 4 waves
// and the partial sums combined through LDS, so a 32 x 32 plane still fills the chip.
__global__ __launch_bounds__(256) void ln_stats_split_kernel(const float* __restrict__ x, long x_bs,
                                                             float* __restrict__ mu_out,
                                                             float* __restrict__ rstd_out, int C, int P) {
  __shared__ float part[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  const bool ok = p < P;
  const float* __restrict__ xp = x + (long)b * x_bs + (ok ? p : 0);
  float s = 0.f;
#pragma unroll 8
  for (int c = wave; c < C; c += 4) s += xp[(long)c * P];
  part[0][wave][lane] = s;
  __syncthreads();
  const float mu = (part[0][0][lane] + part[0][1][lane] + part[0][2][lane] + part[0][3][lane]) / (float)C;
  float v = 0.f;
#pragma unroll 8
  for (int c = wave; c < C; c += 4) {
    const float d = xp[(long)c * P] - mu;
    v = fmaf(d, d, v);
  }
  part[1][wave][lane] = v;
  __syncthreads();
  if (wave == 0 && ok) {
    v = part[1][0][lane] + part[1][1][lane] + part[1][2][lane] + part[1][3][lane];
    mu_out[(long)b * P + p] = mu;
    rstd_out[(long)b * P + p] = 1.0f / sqrtf(v / (float)C + 1e-5f);
  }
}

void launch_ln_stats(TV x, float* mu, float* rstd, int B, int C, int P, hipStream_t s) {
  Scope sc(intern_label("ln_stats C%d P%d", C, P), 5.0 * B * (double)C * P, 4.0 * B * (double)P * (C + 2), s);
  if ((long)B * P <= 16384 && C >= 128 && C <= 1024) {
    const dim3 grid((P + 15) / 16, B), block(256);
    float* const none = nullptr;
    if (C <= 256)
      hipLaunchKernelGGL((layernorm_small_kernel<0, 16>), grid, block, 0, s, x.p, x.bs, none, none, none, 0L, C, P, mu, rstd);
    else if (C <= 512)
      hipLaunchKernelGGL((layernorm_small_kernel<0, 32>), grid, block, 0, s, x.p, x.bs, none, none, none, 0L, C, P, mu, rstd);
    else
      hipLaunchKernelGGL((layernorm_small_kernel<0, 64>), grid, block, 0, s, x.p, x.bs, none, none, none, 0L, C, P, mu, rstd);
  } else if ((long)B * P <= 256 * 1024) {
    hipLaunchKernelGGL(ln_stats_split_kernel, dim3((P + 63) / 64, B), dim3(256), 0, s, x.p, x.bs, mu, rstd, C, P);
  } else {
    hipLaunchKernelGGL(layernorm_kernel<0>, dim3((P + 255) / 256, B), dim3(256), 0, s, x.p, x.bs, nullptr,
                       nullptr, nullptr, 0L, mu, rstd, C, P);
  }
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- depthwise 3x3, zero padding 1 ---------------------------------------------------------------
// A workgroup owns a TH x TW output tile of one channel plane (two planes when a gate is
// fused), staged with its one-pixel halo in LDS; each lane then produces 4 adjacent pixels of a
// row from 3 x 6 LDS values.
//   MODE 0  y[c] = dw(x[c])
//   MODE 1  y[c] = gelu(dw(x[c])) * dw(x[c + Ch])   for c < Ch = C/2   (HINT GDFN)
//   MODE 2  y[c] = dw(x[c]) * x[c + Ch]              (ASTv2 FeedForward: second half un-convolved)
//   MODE 3  y[c] = gelu(dw(x[c])) * x[c + Ch]        (AST FRFN gate)
//   MODE 4  y[c] = gelu(dw(x[c]))                    (AST LeFF: depthwise conv + GELU, no gate)
// wmod > 0 (MODE 0 only) selects the "partial convolution" of ASTv2: channel c uses weight
// c % wmod when c % wmod < wconv and is copied through otherwise.
constexpr int DW_TH = 16, DW_TW = 64;

template <int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, long y_bs, int C,
                                                        int H, int W, int tiles_x, int tiles_y,
                                                        int wmod, int wconv) {
  constexpr int LW = DW_TW + 2 + 2;  // halo + pad so rows are 68 floats (odd multiple of 4 banks)
  constexpr int NP = (MODE == 0 || MODE == 4) ? 1 : 2;
  __shared__ float tile[NP][(DW_TH + 2) * LW];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int c = t / tiles_y;  // output channel
  const int b = blockIdx.y;
  const int x0 = tx * DW_TW, y0 = ty * DW_TH;
  const long P = (long)H * W;
  const int Cout = (MODE == 0 || MODE == 4) ? C : C / 2;

#pragma unroll
  for (int g = 0; g < NP; ++g) {
    const float* __restrict__ plane = x + (long)b * x_bs + (long)(c + g * Cout) * P;
    for (int i = tid; i < (DW_TH + 2) * (DW_TW + 2); i += 256) {
      const int r = i / (DW_TW + 2), q = i - r * (DW_TW + 2);
      const int yy = y0 + r - 1, xx = x0 + q - 1;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      tile[g][r * LW + q] = ok ? plane[(long)yy * W + xx] : 0.f;
    }
  }
  __syncthreads();

  const int r = tid >> 4;          // 0..15 output row in tile
  const int q = (tid & 15) * 4;    // 0..60 output col in tile
  const int oy = y0 + r, ox = x0 + q;
  if (oy >= H || ox >= W) return;
  float res[NP][4];
#pragma unroll
  for (int g = 0; g < NP; ++g) {
    const int ch = c + g * Cout;
    int wch = ch;
    bool through = (MODE == 2 || MODE == 3) && g == 1;
    if (MODE == 0 && wmod > 0) {
      wch = ch % wmod;
      through = wch >= wconv;
    }
    if (through) {
#pragma unroll
      for (int j = 0; j < 4; ++j) res[g][j] = tile[g][(r + 1) * LW + q + 1 + j];
      continue;
    }
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[wch * 9 + i];
    const float bb = bias ? bias[wch] : 0.f;
    float acc[4] = {bb, bb, bb, bb};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      float v[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = tile[g][(r + dy) * LW + q + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = fmaf(k[dy * 3 + 0], v[j], acc[j]);
        acc[j] = fmaf(k[dy * 3 + 1], v[j + 1], acc[j]);
        acc[j] = fmaf(k[dy * 3 + 2], v[j + 2], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) res[g][j] = acc[j];
  }
  float out[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (MODE == 1)
      out[j] = gelu_erf_pw(res[0][j]) * res[NP - 1][j];
    else if constexpr (MODE == 2)
      out[j] = res[0][j] * res[NP - 1][j];
    else if constexpr (MODE == 3)
      out[j] = gelu_erf_pw(res[0][j]) * res[NP - 1][j];
    else if constexpr (MODE == 4)
      out[j] = gelu_erf_pw(res[0][j]);
    else
      out[j] = res[0][j];
  }
  float* __restrict__ dst = y + (long)b * y_bs + (long)c * P + (long)oy * W + ox;
  if (ox + 3 < W && (W & 3) == 0 && (y_bs & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ox + j < W) dst[j] = out[j];
  }
}

// Register-rolling form used whenever rows are float4-addressable (W % 4 == 0, aligned planes):
// a lane owns 4 adjacent pixels and walks down R output rows keeping three input rows in
// registers; the left / right neighbours come from the adjacent lanes (__shfl), only the lanes on
// a tile edge touch memory for them.  One wave per task, no LDS, no barrier; every global access
// is a 16-byte lane access on a contiguous row segment.
template <int MODE, int PX, int RFIX>
__global__ __launch_bounds__(256) void dwconv3x3_rows_kernel(const float* __restrict__ x, long x_bs,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, long y_bs, int C,
                                                             int H, int W, int lx_log2, int R,
                                                             int tiles_x, int strips, int n_tasks,
                                                             int wmod, int wconv) {
  constexpr int NP = (MODE == 0 || MODE == 4) ? 1 : 2;
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= n_tasks) return;  // whole wave
  const int b = blockIdx.y;
  const int LX = 1 << lx_log2, RW = 64 >> lx_log2;
  int t = task;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ts = t % strips;
  const int c = t / strips;
  const int xq = lane & (LX - 1), yr = lane >> lx_log2;
  const int px = (tx * LX + xq) * PX;
  const int ys = (ts * RW + yr) * R;
  const long P = (long)H * W;
  const int Cout = (MODE == 0 || MODE == 4) ? C : C / 2;
  const bool x_ok = px < W;

  struct Row { float v[PX + 2]; };
  struct Raw { float m[PX]; };
  // A row is fetched in two steps so that several rows can be in flight: `fetch` issues the 16-byte (8-byte)
  // load from a clamped, always valid address with no predicate around it; `finish` zeroes what lies outside
  // the image and gathers the left / right neighbours from the adjacent lanes (whole-wave DPP shifts; only
  // the lanes on a tile edge touch memory for them).
  auto fetch = [&](const float* __restrict__ plane, int yy) {
    Raw r;
    const float* __restrict__ p = plane + (long)min(max(yy, 0), H - 1) * W + (x_ok ? px : 0);
    if constexpr (PX == 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(p);
      r.m[0] = t4.x; r.m[1] = t4.y; r.m[2] = t4.z; r.m[3] = t4.w;
    } else {
      const float2 t2 = *reinterpret_cast<const float2*>(p);
      r.m[0] = t2.x; r.m[1] = t2.y;
    }
    return r;
  };
  auto finish = [&](const Raw& raw, const float* __restrict__ plane, int yy) {
    Row r;
    const bool ok = x_ok && yy >= 0 && yy < H;
    float m[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) m[j] = ok ? raw.m[j] : 0.f;
    // lane i takes lane i - 1's last pixel / lane i + 1's first pixel (wave_shr:1 / wave_shl:1)
    float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m[PX - 1]),
                                                                    __builtin_bit_cast(int, m[PX - 1]), 0x138, 0xf, 0xf, false));
    float rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m[0]),
                                                                     __builtin_bit_cast(int, m[0]), 0x130, 0xf, 0xf, false));
    const float* __restrict__ p = plane + (long)yy * W + px;
    if (xq == 0) l = (ok && px > 0) ? p[-1] : 0.f;
    if (xq == LX - 1 || px + PX >= W) rr = (ok && px + PX < W) ? p[PX] : 0.f;
    r.v[0] = l;
#pragma unroll
    for (int j = 0; j < PX; ++j) r.v[1 + j] = m[j];
    r.v[PX + 1] = rr;
    return r;
  };
  auto load_row = [&](const float* __restrict__ plane, int yy) { return finish(fetch(plane, yy), plane, yy); };

  // per-plane state
  float k[NP][9], bb[NP];
  bool conv[NP];
  const float* plane[NP];
#pragma unroll
  for (int g = 0; g < NP; ++g) {
    const int ch = c + g * Cout;
    int wch = ch;
    conv[g] = !((MODE == 2 || MODE == 3) && g == 1);
    if (MODE == 0 && wmod > 0) {
      wch = ch % wmod;
      conv[g] = wch < wconv;
    }
    plane[g] = x + (long)b * x_bs + (long)ch * P;
#pragma unroll
    for (int i = 0; i < 9; ++i) k[g][i] = conv[g] ? w[wch * 9 + i] : 0.f;
    bb[g] = (conv[g] && bias) ? bias[wch] : 0.f;
  }
  float* __restrict__ dst0 = y + (long)b * y_bs + (long)c * P + px;
  // one output row from the three input rows around it
  auto emit = [&](const Row* ra, const Row* rb, const Row* rc, const int oy) {
    float res[NP][PX];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      if (!conv[g]) {
#pragma unroll
        for (int j = 0; j < PX; ++j) res[g][j] = rb[g].v[1 + j];
        continue;
      }
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        float acc = bb[g];
        acc = fmaf(k[g][0], ra[g].v[j], acc);
        acc = fmaf(k[g][1], ra[g].v[j + 1], acc);
        acc = fmaf(k[g][2], ra[g].v[j + 2], acc);
        acc = fmaf(k[g][3], rb[g].v[j], acc);
        acc = fmaf(k[g][4], rb[g].v[j + 1], acc);
        acc = fmaf(k[g][5], rb[g].v[j + 2], acc);
        acc = fmaf(k[g][6], rc[g].v[j], acc);
        acc = fmaf(k[g][7], rc[g].v[j + 1], acc);
        acc = fmaf(k[g][8], rc[g].v[j + 2], acc);
        res[g][j] = acc;
      }
    }
    float o[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      if constexpr (MODE == 1 || MODE == 3) o[j] = gelu_erf_pw(res[0][j]) * res[NP - 1][j];
      else if constexpr (MODE == 2) o[j] = res[0][j] * res[1][j];
      else if constexpr (MODE == 4) o[j] = gelu_erf_pw(res[0][j]);
      else o[j] = res[0][j];
    }
    if (x_ok && oy < H) {
      if constexpr (PX == 4) *reinterpret_cast<float4*>(dst0 + (long)oy * W) = make_float4(o[0], o[1], o[2], o[3]);
      else *reinterpret_cast<float2*>(dst0 + (long)oy * W) = make_float2(o[0], o[1]);
    }
  };
  if constexpr (RFIX > 0) {
    // fixed strip height: every input row of the strip is requested (unpredicated, raw) before the first
    // output row is computed, so a wave keeps (RFIX + 2) row loads per convolved plane in flight; the
    // neighbour gathering happens row by row as the outputs are produced
    constexpr bool PLAIN2F = MODE == 2 || MODE == 3;
    Raw raws[RFIX + 2][NP];
#pragma unroll
    for (int i = 0; i < RFIX + 2; ++i)
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2F && g == 1 && (i == 0 || i == RFIX + 1)) continue;   // the plain plane needs its centre rows only
        raws[i][g] = fetch(plane[g], ys - 1 + i);
      }
    Row ra[NP], rb[NP], rc[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      if (PLAIN2F && g == 1) continue;
      ra[g] = finish(raws[0][g], plane[g], ys - 1);
      rb[g] = finish(raws[1][g], plane[g], ys);
    }
#pragma unroll
    for (int i = 0; i < RFIX; ++i) {
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2F && g == 1) {
          const bool ok = x_ok && ys + i < H;
#pragma unroll
          for (int j = 0; j < PX; ++j) rb[g].v[1 + j] = ok ? raws[i + 1][g].m[j] : 0.f;
        } else {
          rc[g] = finish(raws[i + 2][g], plane[g], ys + i + 1);
        }
      }
      emit(ra, rb, rc, ys + i);
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2F && g == 1) continue;
        ra[g] = rb[g];
        rb[g] = rc[g];
      }
    }
  } else {
    // the second plane of MODE 2 / 3 is multiplied in as it is: only its centre row is read, without
    // the neighbours
    constexpr bool PLAIN2 = MODE == 2 || MODE == 3;
    auto load_center = [&](const float* __restrict__ pl, int yy) {
      Row r;
#pragma unroll
      for (int j = 0; j < PX + 2; ++j) r.v[j] = 0.f;
      if (x_ok && yy < H) {
        const float* __restrict__ p = pl + (long)yy * W + px;
        if constexpr (PX == 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(p);
          r.v[1] = t4.x; r.v[2] = t4.y; r.v[3] = t4.z; r.v[4] = t4.w;
        } else {
          const float2 t2 = *reinterpret_cast<const float2*>(p);
          r.v[1] = t2.x; r.v[2] = t2.y;
        }
      }
      return r;
    };
    Row ra[NP], rb[NP], rc[NP];
    Raw nx[NP];   // the next input row of every convolved plane, in flight while the current one is used
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      if (PLAIN2 && g == 1) continue;
      const Raw r0 = fetch(plane[g], ys - 1), r1 = fetch(plane[g], ys);
      nx[g] = fetch(plane[g], ys + 1);
      ra[g] = finish(r0, plane[g], ys - 1);
      rb[g] = finish(r1, plane[g], ys);
    }
    for (int i = 0; i < R; ++i) {
      const int oy = ys + i;
      Raw cur[NP];
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2 && g == 1) continue;
        cur[g] = nx[g];
        nx[g] = fetch(plane[g], i + 1 < R ? oy + 2 : oy + 1);   // last trip: a row already cached, not a new one
      }
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2 && g == 1) rb[g] = load_center(plane[g], oy);
        else rc[g] = finish(cur[g], plane[g], oy + 1);
      }
      emit(ra, rb, rc, oy);
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        if (PLAIN2 && g == 1) continue;
        ra[g] = rb[g];
        rb[g] = rc[g];
      }
    }
  }
}

void launch_dwconv3x3(TV x, const float* w, const float* bias, TV y, int B, int C, int H, int W,
                      int gate, hipStream_t s, int wmod, int wconv) {
  JIRF_REQUIRE(gate >= 0 && gate <= 4, JIRF_EINVAL, "dwconv3x3: mode %d", gate);
  JIRF_REQUIRE(gate == 0 || gate == 4 || C % 2 == 0, JIRF_EINVAL, "dwconv3x3: gated forms need an even channel count");
  JIRF_REQUIRE(wmod == 0 || (gate == 0 && wconv > 0 && wconv <= wmod && C % wmod == 0), JIRF_EINVAL,
               "dwconv3x3: bad partial-convolution pattern %d/%d for C=%d", wconv, wmod, C);
  const int tiles_x = (W + DW_TW - 1) / DW_TW, tiles_y = (H + DW_TH - 1) / DW_TH;
  const int Cout = (gate && gate != 4) ? C / 2 : C;
  const dim3 grid((unsigned)(tiles_x * tiles_y * Cout), B), block(256);
  const double px = (double)B * H * W;
  const char* nm = wmod ? "dwconv3x3_partial" : gate == 1 ? "dwconv3x3_gate" : gate == 2 ? "dwconv3x3_mul"
                   : gate == 3 ? "dwconv3x3_gelu_mul" : gate == 4 ? "dwconv3x3_gelu" : "dwconv3x3";
  Scope sc(intern_label("%s C%d H%d W%d", nm, C, H, W), 18.0 * px * ((gate == 2 || gate == 3) ? C / 2 : C), 4.0 * px * (C + Cout), s);
  const uintptr_t addr = reinterpret_cast<uintptr_t>(x.p) | reinterpret_cast<uintptr_t>(y.p);
  const int PX = (W % 4 == 0 && x.bs % 4 == 0 && y.bs % 4 == 0 && (addr & 15) == 0)   ? 4
                 : (W % 2 == 0 && x.bs % 2 == 0 && y.bs % 2 == 0 && (addr & 7) == 0) ? 2
                                                                                     : 0;
  if (PX) {
    int lx_log2 = 0;
    while ((1 << lx_log2) < 64 && (PX << lx_log2) < W) ++lx_log2;   // lanes across a row: up to 64 x PX px
    const int LX = 1 << lx_log2, RW = 64 >> lx_log2;
    int R = std::max(1, std::min(8, H / RW));
    // under two waves per SIMD in the whole grid (small planes of a single image): half-height strips, twice the waves
    // (C1020 64 x 64 at B = 1: 13.5 -> 10.6 us)
    if ((long)((W + PX * LX - 1) / (PX * LX)) * ((H + RW * R - 1) / (RW * R)) * Cout * B < 2048) R = std::min(R, 4);
    const int tx = (W + PX * LX - 1) / (PX * LX), strips = (H + RW * R - 1) / (RW * R);
    const int n_tasks = tx * strips * Cout;
    const dim3 g2((unsigned)((n_tasks + 3) / 4), B);
#define JIRF_DW_ROWS(M, WM, WC)                                                                                      \
  do {                                                                                                               \
    if (PX == 4 && R == 8)                                                                                                  \
      hipLaunchKernelGGL((dwconv3x3_rows_kernel<M, 4, 8>), g2, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,  \
                         lx_log2, R, tx, strips, n_tasks, WM, WC);                                                   \
    else if (PX == 4)                                                                                                \
      hipLaunchKernelGGL((dwconv3x3_rows_kernel<M, 4, 0>), g2, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,  \
                         lx_log2, R, tx, strips, n_tasks, WM, WC);                                                   \
    else                                                                                                             \
      hipLaunchKernelGGL((dwconv3x3_rows_kernel<M, 2, 0>), g2, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,  \
                         lx_log2, R, tx, strips, n_tasks, WM, WC);                                                   \
  } while (0)
    if (gate == 1) JIRF_DW_ROWS(1, 0, 0);
    else if (gate == 2) JIRF_DW_ROWS(2, 0, 0);
    else if (gate == 3) JIRF_DW_ROWS(3, 0, 0);
    else if (gate == 4) JIRF_DW_ROWS(4, 0, 0);
    else JIRF_DW_ROWS(0, wmod, wconv);
#undef JIRF_DW_ROWS
    JIRF_HIP_CHECK(hipGetLastError());
    return;
  }
  if (gate == 1)
    hipLaunchKernelGGL(dwconv3x3_kernel<1>, grid, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,
                       tiles_x, tiles_y, 0, 0);
  else if (gate == 2)
    hipLaunchKernelGGL(dwconv3x3_kernel<2>, grid, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,
                       tiles_x, tiles_y, 0, 0);
  else if (gate == 3)
    hipLaunchKernelGGL(dwconv3x3_kernel<3>, grid, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,
                       tiles_x, tiles_y, 0, 0);
  else if (gate == 4)
    hipLaunchKernelGGL(dwconv3x3_kernel<4>, grid, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,
                       tiles_x, tiles_y, 0, 0);
  else
    hipLaunchKernelGGL(dwconv3x3_kernel<0>, grid, block, 0, s, x.p, x.bs, w, bias, y.p, y.bs, C, H, W,
                       tiles_x, tiles_y, wmod, wconv);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- dense 3x3 with a handful of output channels (the RGB output convolutions) ------------------
// As a GEMM this shape wastes 61 of 64 tile rows.  Here a lane owns 4 adjacent pixels and R rows and
// keeps COUT x R x 4 accumulators; per input channel it loads R + 2 row segments (float4 + the two
// neighbours from the adjacent lanes) and the 9 x COUT weights arrive as scalar loads.
// CSPLIT: a single small image gives the chip one wave per CU, each walking all input channels one
// load round trip after the other; then the four waves of a workgroup share one tile, wave w taking
// channels w, w + 4, ..., and their partial sums meet in LDS in wave order (fixed order: reproducible).
template <int COUT, int R, bool CSPLIT = false>
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const float* __restrict__ x, long x_bs,
                                                             const float* __restrict__ wt, int Mpad,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ res, long r_bs,
                                                             float* __restrict__ y, long y_bs, int Cin,
                                                             int H, int W, int lx_log2, int tiles_x,
                                                             int n_tasks) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int task = CSPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (task >= n_tasks) return;  // whole wave (CSPLIT: whole workgroup)
  const int b = blockIdx.y;
  const int LX = 1 << lx_log2, RW = 64 >> lx_log2;
  const int tx = task % tiles_x, ts = task / tiles_x;
  const int xq = lane & (LX - 1), yr = lane >> lx_log2;
  const int px = (tx * LX + xq) * 4;
  const int ys = (ts * RW + yr) * R;
  const long P = (long)H * W;
  const bool x_ok = px < W;
  float acc[COUT][R][4];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[o][i][j] = 0.f;
  const float* __restrict__ xb = x + (long)b * x_bs;
  for (int c = CSPLIT ? wave : 0; c < Cin; c += CSPLIT ? 4 : 1) {
    const float* __restrict__ plane = xb + (long)c * P;
    float wv[9][COUT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < COUT; ++o) wv[t][o] = wt[(long)(t * Cin + c) * Mpad + o];
    float row[R + 2][6];
#pragma unroll
    for (int i = 0; i < R + 2; ++i) {
      const int yy = ys + i - 1;
      const bool ok = x_ok && yy >= 0 && yy < H;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* __restrict__ p = plane + (long)yy * W + px;
      if (ok) m = *reinterpret_cast<const float4*>(p);
      float l = __shfl_up(m.w, 1), rr = __shfl_down(m.x, 1);
      if (xq == 0) l = (ok && px > 0) ? p[-1] : 0.f;
      if (xq == LX - 1 || px + 4 >= W) rr = (ok && px + 4 < W) ? p[4] : 0.f;
      row[i][0] = l; row[i][1] = m.x; row[i][2] = m.y; row[i][3] = m.z; row[i][4] = m.w; row[i][5] = rr;
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < COUT; ++o)
              acc[o][i][j] = fmaf(wv[ky * 3 + kx][o], row[i + ky][j + kx], acc[o][i][j]);
  }
  if constexpr (CSPLIT) {
    __shared__ float part[3][COUT * R * 4][64];
    if (wave > 0) {
#pragma unroll
      for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) part[wave - 1][(o * R + i) * 4 + j][lane] = acc[o][i][j];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int o = 0; o < COUT; ++o)
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[o][i][j] += part[w][(o * R + i) * 4 + j][lane];
  }
  if (!x_ok) return;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float bo = bias ? bias[o] : 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int oy = ys + i;
      if (oy >= H) continue;
      const long off = (long)o * P + (long)oy * W + px;
      float4 v = make_float4(acc[o][i][0] + bo, acc[o][i][1] + bo, acc[o][i][2] + bo, acc[o][i][3] + bo);
      if (res) {
        const float4 rv = *reinterpret_cast<const float4*>(res + (long)b * r_bs + off);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      *reinterpret_cast<float4*>(y + (long)b * y_bs + off) = v;
    }
  }
}

bool try_launch_conv3x3_fewout(const GemmArgs& a, hipStream_t s) {
  auto al = [](const float* p, long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && bs % 4 == 0; };
  // up to 4 output channels with 4 rows per lane; 8 with 2 rows, 16 with 1 row (the quarter-width dense
  // 3x3 of AST's FRFN at its two finest levels): 64 accumulators per lane in every case
  const bool m_ok = (a.M >= 1 && a.M <= 4) || a.M == 8 || a.M == 16;
  const bool shape_ok = m_ok && a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Hin == a.H &&
                        a.Win == a.W && a.W % 4 == 0 && (a.epi == EP_STORE || a.epi == EP_RESID) && a.wt_bs == 0 &&
                        al(a.x.p, a.x.bs) && al(a.y.p, a.y.bs) && (a.epi != EP_RESID || al(a.r.p, a.r.bs));
  if (!shape_ok) return false;
  const int H = a.H, W = a.W;
  int lx_log2 = 0;
  while ((1 << lx_log2) < 64 && (4 << lx_log2) < W) ++lx_log2;
  const int LX = 1 << lx_log2, RW = 64 >> lx_log2;
  // rows per lane: as many as the accumulators allow, fewer while that leaves under 512 waves in the grid
  int R = a.M <= 4 ? 4 : a.M == 8 ? 2 : 1;
  while (R > 1 && (long)((W + 4 * LX - 1) / (4 * LX)) * ((H + RW * R - 1) / (RW * R)) * a.B < 512) R >>= 1;
  const int tiles_x = (W + 4 * LX - 1) / (4 * LX), strips = (H + RW * R - 1) / (RW * R);
  const int n_tasks = tiles_x * strips;
  // 8 / 16 output channels: one wave carries 128-256 pixels x all channels x all taps; with under ~1000 waves in
  // the grid (single images) the matrix-core GEMM on a padded tile is faster (AST 256^2, B = 1: 70 -> ~25 us)
  if (a.M >= 8 && (long)n_tasks * a.B < 1024) return false;
  const dim3 grid((unsigned)((n_tasks + 3) / 4), a.B), block(256);
  const float* res = a.epi == EP_RESID ? a.r.p : nullptr;
  // a single small image (R is already 1): the input channels are split over the four waves of a workgroup
  if (a.M <= 4 && R == 1 && (long)n_tasks * a.B < 512 && a.Cin >= 16) {
    const dim3 grid1((unsigned)n_tasks, a.B);
#define JIRF_FEWOUT_CS(CO)                                                                                     \
  hipLaunchKernelGGL((conv3x3_fewout_kernel<CO, 1, true>), grid1, block, 0, s, a.x.p, a.x.bs, a.wt, a.Mpad, a.bias, \
                     res, a.r.bs, a.y.p, a.y.bs, a.Cin, H, W, lx_log2, tiles_x, n_tasks)
    switch (a.M) {
      case 1: JIRF_FEWOUT_CS(1); break;
      case 2: JIRF_FEWOUT_CS(2); break;
      case 3: JIRF_FEWOUT_CS(3); break;
      default: JIRF_FEWOUT_CS(4); break;
    }
#undef JIRF_FEWOUT_CS
    JIRF_HIP_CHECK(hipGetLastError());
    return true;
  }
#define JIRF_FEWOUT(CO, RR)                                                                            \
  hipLaunchKernelGGL((conv3x3_fewout_kernel<CO, RR>), grid, block, 0, s, a.x.p, a.x.bs, a.wt, a.Mpad, a.bias, res, \
                     a.r.bs, a.y.p, a.y.bs, a.Cin, H, W, lx_log2, tiles_x, n_tasks)
#define JIRF_FEWOUT_R(CO)              \
  do {                                 \
    if (R == 4) JIRF_FEWOUT(CO, 4);    \
    else if (R == 2) JIRF_FEWOUT(CO, 2); \
    else JIRF_FEWOUT(CO, 1);           \
  } while (0)
  switch (a.M) {
    case 1: JIRF_FEWOUT_R(1); break;
    case 2: JIRF_FEWOUT_R(2); break;
    case 3: JIRF_FEWOUT_R(3); break;
    case 4: JIRF_FEWOUT_R(4); break;
    case 8:
      if (R == 2) JIRF_FEWOUT(8, 2);
      else JIRF_FEWOUT(8, 1);
      break;
    default: JIRF_FEWOUT(16, 1); break;
  }
#undef JIRF_FEWOUT_R
#undef JIRF_FEWOUT
  JIRF_HIP_CHECK(hipGetLastError());
  return true;
}

// ---- accuracy scan of the branch-free GELU (diagnostics) ------------------------------------------------------
// Every float u in [0, hi] by bit pattern, both signs: the float64 value of erfc(u / sqrt 2) against 2 half_erfc_abs(u),
// and gelu_fast(+-u) against 0.5 v (1 + erf(v / sqrt 2)) in float64 and against the same form on the C library's erff
// (the previous implementation).  out [grid][4] maxima per workgroup: |erfc error|, |gelu error|, gelu error in units
// of the float spacing at the exact value (v >= 0 only: for v < 0 gelu decays to zero like erfc and the absolute error
// is the measure), |gelu_fast - erff form|.
__global__ __launch_bounds__(256) void gelu_scan_kernel(unsigned last_bits, double* __restrict__ part) {
  __shared__ double red[4][256];
  double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i <= last_bits;
       i += (unsigned long long)gridDim.x * 256) {
    const float u = __uint_as_float((unsigned)i);
    const double ud = (double)u;
    const double ec = erfc(ud * 0.70710678118654752440);
    m0 = fmax(m0, fabs(2.0 * (double)half_erfc_abs(u) - ec));
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
      const float v = sgn ? -u : u;
      const double want = sgn ? -0.5 * ud * ec : 0.5 * ud * (2.0 - ec);
      const float got = gelu_fast(v);
      const double e = fabs((double)got - want);
      m1 = fmax(m1, e);
      const float wf = (float)want;
      const float ulp = fabsf(__uint_as_float(__float_as_uint(fabsf(wf)) + 1u) - fabsf(wf));
      if (!sgn && fabsf(wf) > 1e-30f) m2 = fmax(m2, e / (double)ulp);   // v < 0: gelu -> 0-, only the absolute error means anything
      const float old = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      m3 = fmax(m3, fabs((double)got - (double)old));
    }
  }
  red[0][threadIdx.x] = m0; red[1][threadIdx.x] = m1; red[2][threadIdx.x] = m2; red[3][threadIdx.x] = m3;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = fmax(red[j][threadIdx.x], red[j][threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

void gelu_error_scan(float hi, double out[4], hipStream_t s) {
  JIRF_REQUIRE(hi > 0.f && hi <= 3e38f, JIRF_EINVAL, "gelu scan: bad upper bound %g", (double)hi);
  unsigned bits;
  std::memcpy(&bits, &hi, 4);
  const int grid = 4096;
  double* part = nullptr;
  JIRF_HIP_CHECK(hipMalloc(&part, (size_t)grid * 4 * sizeof(double)));
  hipLaunchKernelGGL(gelu_scan_kernel, dim3(grid), dim3(256), 0, s, bits, part);
  std::vector<double> host((size_t)grid * 4);
  const hipError_t e1 = hipGetLastError();
  const hipError_t e2 = hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
  const hipError_t e3 = hipStreamSynchronize(s);
  (void)hipFree(part);
  JIRF_HIP_CHECK(e1);
  JIRF_HIP_CHECK(e2);
  JIRF_HIP_CHECK(e3);
  for (int j = 0; j < 4; ++j) out[j] = 0.0;
  for (int b = 0; b < grid; ++b)
    for (int j = 0; j < 4; ++j) out[j] = std::max(out[j], host[(size_t)b * 4 + j]);
}

}  // namespace jirf
