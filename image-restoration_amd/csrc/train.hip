// Backward and optimiser kernels for the data-parallel training step of AST
// (reference basicsr/models/image_restoration_model.py:235-276: forward, smooth-L1 pixel loss
// basicsr/models/losses/losses.py:12-14, backward, gradient-norm clip 0.01, Adam; the gradient average
// over ranks happens inside Jittor's optimizer, basicsr/models/base_model.py:287-305 reduces the logged
// loss).  This file holds the device side of that step as far as it is built: the loss, the fused
// clip + Adam update on flat parameter / gradient buffers, and the backward of the operators that carry
// most of AST's parameters and FLOPs — Linear / 1x1 convolution, LayerNorm, the GELU gates and the
// depthwise 3x3 of the FRFN token MLP.  Everything is fp32, planar [B][C][P] like the forward, and every
// reduction is two-stage with a fixed order (no atomics), so a step is reproducible.
#include <algorithm>
#include <cmath>

#include "runtime.h"

namespace jirf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// d/dv [0.5 v (1 + erf(v / sqrt 2))] = 0.5 (1 + erf(v / sqrt 2)) + v exp(-v^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_grad_f(float v) {
  return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * expf(-0.5f * v * v) * 0.39894228040143267794f;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// ---- loss ---------------------------------------------------------------------------------------------
// nn.smooth_l1_loss(pred, target, reduction="none") then mean (losses.py:12-14, loss_util weighted_loss):
//   l = 0.5 d^2 for |d| < 1, |d| - 0.5 otherwise;  dl/dpred = clamp(d, -1, 1);  both scaled by weight / n.
// part [gridDim.x] doubles; the caller sums them in order.
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        float* __restrict__ grad, long n, float scale,
                                                        double* __restrict__ part) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = pred[i] - tgt[i];
    const float ad = fabsf(d);
    acc += ad < 1.0f ? 0.5 * (double)d * d : (double)ad - 0.5;
    if (grad) grad[i] = fminf(fmaxf(d, -1.0f), 1.0f) * scale;
  }
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// Sum of squares of a flat buffer (the gradient norm of clip_grad_norm): part [gridDim.x].
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, double* __restrict__ part) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)x[i] * x[i];
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void sum_parts_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < n; ++i) t += part[i];
    out[0] = t;
  }
}

// Gradient clip + Adam on flat buffers, Jittor's update (jt.optim.Adam.step):
//   g <- g * min(max_norm / (||g|| + 1e-6), 1)                     (Optimizer.clip_grad_norm; max_norm <= 0: off)
//   m <- b0 m + (1 - b0) g ;  v <- b1 v + (1 - b1) g^2
//   p <- p - m * lr * sqrt(1 - b1^t) / (1 - b0^t) / (sqrt(v) + eps)      (eps outside the bias correction)
// torch_form != 0 selects torch's placement instead: p <- p - lr / (1 - b0^t) * m / (sqrt(v) / sqrt(1 - b1^t) + eps).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const double* __restrict__ sumsq, float max_norm, float lr, float b0,
                                                   float b1, float eps, float bc0, float bc1, int torch_form) {
  float clip = 1.0f;
  if (max_norm > 0.f && sumsq) clip = fminf(max_norm / ((float)sqrt(sumsq[0]) + 1e-6f), 1.0f);
  const float step = lr * sqrtf(bc1) / bc0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * clip;
    const float mi = b0 * m[i] + (1.0f - b0) * gi;
    const float vi = b1 * v[i] + (1.0f - b1) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= torch_form ? (lr / bc0) * mi / (sqrtf(vi) / sqrtf(bc1) + eps) : mi * step / (sqrtf(vi) + eps);
  }
}

// ---- weight gradient of a Linear / 1x1 convolution ----------------------------------------------------------
// dW[o][k] = sum_{b,p} dY[b][o][p] X[b][k][p]: a contraction over the pixel axis, which both operands have
// as their contiguous axis.  A workgroup owns a 64 x 64 tile of dW and one slab of pixels; each of its four
// waves walks over 16-pixel chunks of the slab: the two [64][16] operand chunks go through the wave's own
// LDS region (rows padded to 17 floats, so that reading one pixel column of 32 rows is conflict-free),
// 8 k-steps of v_mfma_f32_32x32x2_f32 on a 2 x 2 tile grid follow; no workgroup barrier in the loop.  The
// four waves' sums are added through LDS in wave order and written to slab [blockIdx.z]; a second kernel
// adds the slabs in order.  db[o] = sum dY rides along (k-tile 0 only).
constexpr int NT_T = 64, NT_PX = 16, NT_LD = NT_PX + 1;
constexpr int NT_LPR = NT_PX / 4, NT_RPI = 64 / NT_LPR, NT_U = NT_T / NT_RPI;   // loader: lanes per row, rows per instruction

__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const float* __restrict__ dy, long dy_bs,
                                                           const float* __restrict__ x, long x_bs,
                                                           float* __restrict__ part, double* __restrict__ dbpart,
                                                           int B, int M, int K, int P, int chunks_per_slab) {
  __shared__ float lds[4][2][NT_T * NT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * NT_T, k0 = blockIdx.y * NT_T, slab = blockIdx.z;
  const int chunks_img = (P + NT_PX - 1) / NT_PX;
  const long chunks = (long)chunks_img * B;
  const long c_lo = (long)slab * chunks_per_slab, c_hi = min(chunks, c_lo + chunks_per_slab);
  float* __restrict__ A = lds[wave][0];   // dY chunk [64][33]
  float* __restrict__ Bm = lds[wave][1];  // X chunk  [64][33]
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  double dbacc = 0.0;   // lane: row m0 + lane, over this wave's chunks
  const int lr = lane / NT_LPR, lc = (lane % NT_LPR) * 4;   // loader: NT_RPI rows x NT_LPR float4 per instruction
  for (long c = c_lo + wave; c < c_hi; c += 4) {
    const int b = (int)(c / chunks_img);
    const int p0 = (int)(c - (long)b * chunks_img) * NT_PX;
    const float* __restrict__ dyb = dy + (long)b * dy_bs;
    const float* __restrict__ xb = x + (long)b * x_bs;
    float4 va[NT_U], vb[NT_U];
#pragma unroll
    for (int u = 0; u < NT_U; ++u) {
      const int row = lr + NT_RPI * u;
      const int pa = min(p0 + lc, P - 4);   // clamped address; the tail is zeroed below
      va[u] = *reinterpret_cast<const float4*>(dyb + (long)min(m0 + row, M - 1) * P + pa);
      vb[u] = *reinterpret_cast<const float4*>(xb + (long)min(k0 + row, K - 1) * P + pa);
    }
#pragma unroll
    for (int u = 0; u < NT_U; ++u) {
      const int row = lr + NT_RPI * u;
      const bool oka = m0 + row < M, okb = k0 + row < K;
      float a4[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, b4[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool pin = p0 + lc + j < P && p0 + lc == min(p0 + lc, P - 4);
        A[row * NT_LD + lc + j] = (oka && pin) ? a4[j] : 0.f;
        Bm[row * NT_LD + lc + j] = (okb && pin) ? b4[j] : 0.f;
      }
    }
    // (the wave's own region: its writes are visible to its reads after the LDS counter wait)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    if (blockIdx.y == 0) {
      float s = 0.f;
#pragma unroll 8
      for (int j = 0; j < NT_PX; ++j) s += A[lane * NT_LD + j];
      dbacc += s;
    }
#pragma unroll
    for (int s = 0; s < NT_PX / 2; ++s) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = A[(i * 32 + l31) * NT_LD + 2 * s + half];
        bf[i] = Bm[(i * 32 + l31) * NT_LD + 2 * s + half];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- the four waves' tiles, added in wave order through LDS (reusing the operand regions) ----------------
  __syncthreads();
  float* __restrict__ red = &lds[0][0][0];          // [64][64] floats = 16 KiB of the 34 KiB
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = j * 32 + l31;
            if (w == 0) red[row * 64 + col] = acc[i][j][r];
            else red[row * 64 + col] += acc[i][j][r];
          }
    }
    __syncthreads();
  }
  float* __restrict__ out = part + (long)slab * M * K;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int row = e >> 6, col = e & 63;
    if (m0 + row < M && k0 + col < K) out[(long)(m0 + row) * K + k0 + col] = red[e];
  }
  if (blockIdx.y == 0 && dbpart) {
    __shared__ double dbs[4][64];
    dbs[wave][lane] = dbacc;
    __syncthreads();
    if (tid < 64 && m0 + tid < M) dbpart[(long)slab * M + m0 + tid] = dbs[0][tid] + dbs[1][tid] + dbs[2][tid] + dbs[3][tid];
  }
}

// out[e] (+)= sum over slabs, in slab order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, long n, int slabs,
                                                           float* __restrict__ out, int accumulate) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float t = 0.f;
  for (int s = 0; s < slabs; ++s) t += part[(long)s * n + e];
  out[e] = accumulate ? out[e] + t : t;
}
__global__ __launch_bounds__(256) void reduce_parts_d_kernel(const double* __restrict__ part, long n, int slabs,
                                                             float* __restrict__ out, int accumulate) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  double t = 0.0;
  for (int s = 0; s < slabs; ++s) t += part[(long)s * n + e];
  out[e] = accumulate ? out[e] + (float)t : (float)t;
}

// ---- LayerNorm over channels, backward ----------------------------------------------------------------
// n = (x - mu) rs (WithBias) or x rs (BiasFree, hint/astv2; AST's nn.LayerNorm is the WithBias form);
// y = n w + b.  With g = dy w per channel:
//   WithBias:  dx = rs (g - mean_c(g) - nhat mean_c(g nhat)),  nhat = (x - mu) rs
//   BiasFree:  dx = rs g - (x - mu) rs^3 mean_c(g x)          (the variance still centres x)
// A thread owns one pixel and loops over the channels twice (loads along pixels are coalesced).
// dw / db partial sums per workgroup: part_w / part_b [gridDim.x * B... ][C], reduced by reduce_parts_d_kernel.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long x_bs,
                                                            const float* __restrict__ dy, long dy_bs,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            long dx_bs, double* __restrict__ part_w,
                                                            double* __restrict__ part_b, int C, int P, int with_bias) {
  extern __shared__ double ln_red[];   // [256]
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool on = p < P;
  const float* __restrict__ xb = x + (long)b * x_bs + (on ? p : 0);
  const float* __restrict__ gb = dy + (long)b * dy_bs + (on ? p : 0);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += xb[(long)c * P];
  const float mu = sum / (float)C;
  float var = 0.f;
  for (int c = 0; c < C; ++c) {
    const float d = xb[(long)c * P] - mu;
    var = fmaf(d, d, var);
  }
  const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
  float s1 = 0.f, s2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gb[(long)c * P] * w[c];
    const float xv = xb[(long)c * P];
    s1 += g;
    s2 += with_bias ? g * (xv - mu) * rs : g * xv;
  }
  s1 /= (float)C;
  s2 /= (float)C;
  float* __restrict__ dxb = dx + (long)b * dx_bs + (on ? p : 0);
  const long slot = ((long)blockIdx.y * gridDim.x + blockIdx.x) * C;
  for (int c = 0; c < C; ++c) {
    const float xv = xb[(long)c * P], dyv = gb[(long)c * P];
    const float g = dyv * w[c];
    const float nh = (xv - mu) * rs;
    float o;
    if (with_bias) o = rs * (g - s1 - nh * s2);
    else o = rs * g - (xv - mu) * rs * rs * rs * s2;
    if (on) dxb[(long)c * P] = o;
    // parameter gradients: dw[c] = sum dy n, db[c] = sum dy  (n = nhat or x rs)
    const double tw = block_sum_d(on ? (double)dyv * (with_bias ? nh : xv * rs) : 0.0, ln_red);
    const double tb = block_sum_d(on ? (double)dyv : 0.0, ln_red);
    if (threadIdx.x == 0) {
      part_w[slot + c] = tw;
      part_b[slot + c] = tb;
    }
  }
}

// ---- elementwise backward pieces of the token MLPs ---------------------------------------------------------
// mode 0: dx = dy * gelu'(pre)                                  (Linear + GELU: linear1 of FRFN / LeFF / Mlp)
// mode 1: FRFN gate y = gelu(a) * b with a = pre [first h channels], b = x2 [h channels]:
//         da = dy * b * gelu'(a)  -> dx[0:h],   db = dy * gelu(a)  -> dx[h:2h]
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ x2,
                                                       const float* __restrict__ dy, float* __restrict__ dx, long n_img,
                                                       long pre_bs, long x2_bs, long dy_bs, long dx_bs, long half_off,
                                                       int mode) {
  const int b = blockIdx.y;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_img; i += (long)gridDim.x * 256) {
    const float a = pre[(long)b * pre_bs + i], g = dy[(long)b * dy_bs + i];
    if (mode == 0) {
      dx[(long)b * dx_bs + i] = g * gelu_grad_f(a);
    } else {
      const float bv = x2[(long)b * x2_bs + i];
      dx[(long)b * dx_bs + i] = g * bv * gelu_grad_f(a);
      dx[(long)b * dx_bs + half_off + i] = g * gelu_f(a);
    }
  }
}

// ---- depthwise 3x3, weight and bias gradients -------------------------------------------------------------
// dw[c][ky][kx] = sum_{b,y,x} dy[b][c][y][x] * x[b][c][y + ky - 1][x + kx - 1] (zero padded), db[c] = sum dy.
// One workgroup per (channel, image): nine + one sums in double; part [B][C][10], reduced in image order.
__global__ __launch_bounds__(256) void dwconv_dw_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ dy, long dy_bs,
                                                        double* __restrict__ part, int C, int H, int W) {
  __shared__ double red[256];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* __restrict__ xp = x + (long)b * x_bs + (long)c * H * W;
  const float* __restrict__ gp = dy + (long)b * dy_bs + (long)c * H * W;
  double acc[10];
  for (int t = 0; t < 10; ++t) acc[t] = 0.0;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int yy = i / W, xx = i - yy * W;
    const float g = gp[i];
    acc[9] += g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sy = yy + ky - 1, sx = xx + kx - 1;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc[ky * 3 + kx] += (double)g * xp[sy * W + sx];
      }
  }
  for (int t = 0; t < 10; ++t) {
    const double v = block_sum_d(acc[t], red);
    if (threadIdx.x == 0) part[((long)b * C + c) * 10 + t] = v;
  }
}
__global__ __launch_bounds__(256) void dwconv_dw_reduce_kernel(const double* __restrict__ part, int B, int C,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= C * 10) return;
  const int c = e / 10, t = e - c * 10;
  double s = 0.0;
  for (int b = 0; b < B; ++b) s += part[((long)b * C + c) * 10 + t];
  float* dst = t < 9 ? dw + c * 9 + t : (db ? db + c : nullptr);
  if (dst) *dst = accumulate ? *dst + (float)s : (float)s;
}

template <typename F>
int guarded_t(F&& f) {
  try {
    return f();
  } catch (const Failure& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    set_error("host allocation failed");
    return JIRF_ENOMEM;
  }
}

}  // namespace
}  // namespace jirf

using namespace jirf;

extern "C" {

int jirf_train_smooth_l1(const float* pred, const float* target, float* grad, long n, float loss_weight,
                         double* loss_host, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(pred && target && n > 0 && loss_host, JIRF_EINVAL, "smooth_l1: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (int)std::min<long>(1024, (n + 255) / 256);
    DeviceBuffer part;
    part.ensure((size_t)(blocks + 1) * sizeof(double));
    double* pd = static_cast<double*>(part.p);
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(blocks), dim3(256), 0, s, pred, target, grad, n,
                       loss_weight / (float)n, pd);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, s, pd, blocks, pd + blocks);
    JIRF_HIP_CHECK(hipGetLastError());
    double tot = 0.0;
    JIRF_HIP_CHECK(hipMemcpyAsync(&tot, pd + blocks, sizeof(double), hipMemcpyDeviceToHost, s));
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    *loss_host = tot * (double)loss_weight / (double)n;
    return JIRF_OK;
  });
}

int jirf_train_sumsq(const float* x, long n, double* out_dev, double* out_host, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && n > 0 && (out_dev || out_host), JIRF_EINVAL, "sumsq: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (int)std::min<long>(1024, (n + 255) / 256);
    DeviceBuffer part;
    part.ensure((size_t)(blocks + 1) * sizeof(double));
    double* pd = static_cast<double*>(part.p);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, x, n, pd);
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, s, pd, blocks, out_dev ? out_dev : pd + blocks);
    JIRF_HIP_CHECK(hipGetLastError());
    if (out_host)
      JIRF_HIP_CHECK(hipMemcpyAsync(out_host, out_dev ? out_dev : pd + blocks, sizeof(double), hipMemcpyDeviceToHost, s));
    JIRF_HIP_CHECK(hipStreamSynchronize(s));   // `part` is released on return
    return JIRF_OK;
  });
}

int jirf_train_adam_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq_dev, float max_norm,
                         float lr, float beta0, float beta1, float eps, int step, int torch_form, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(p && g && m && v && n > 0 && step >= 1, JIRF_EINVAL, "adam_step: bad arguments");
    JIRF_REQUIRE(max_norm <= 0.f || sumsq_dev, JIRF_EINVAL, "adam_step: clipping needs the gradient's sum of squares");
    const float bc0 = 1.0f - std::pow(beta0, (float)step), bc1 = 1.0f - std::pow(beta1, (float)step);
    const int blocks = (int)std::min<long>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                       sumsq_dev, max_norm, lr, beta0, beta1, eps, bc0, bc1, torch_form);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int B, int M, int K, int P,
                                 int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(dy && x && dw && B > 0 && M > 0 && K > 0 && P >= 4 && P % 4 == 0, JIRF_EINVAL,
                 "linear_bwd_weight: bad arguments (P must be a multiple of 4)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tm = (M + NT_T - 1) / NT_T, tk = (K + NT_T - 1) / NT_T;
    const long chunks = (long)((P + NT_PX - 1) / NT_PX) * B;
    // enough slabs to fill the chip, at least 16 chunks (4 per wave) each
    int slabs = (int)std::max<long>(1, std::min<long>(chunks / 16, (1024 + tm * tk - 1) / (tm * tk)));
    const int cps = (int)((chunks + slabs - 1) / slabs);
    slabs = (int)((chunks + cps - 1) / cps);
    DeviceBuffer part, dbpart;
    part.ensure((size_t)slabs * M * K * sizeof(float));
    if (db) dbpart.ensure((size_t)slabs * M * sizeof(double));
    hipLaunchKernelGGL(linear_dw_kernel, dim3(tm, tk, slabs), dim3(256), 0, s, dy, (long)M * P, x, (long)K * P,
                       static_cast<float*>(part.p), db ? static_cast<double*>(dbpart.p) : nullptr, B, M, K, P, cps);
    const long n = (long)M * K;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float*>(part.p), n, slabs, dw, accumulate);
    if (db)
      hipLaunchKernelGGL(reduce_parts_d_kernel, dim3((M + 255) / 256), dim3(256), 0, s,
                         static_cast<const double*>(dbpart.p), (long)M, slabs, db, accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_train_layernorm_bwd(const float* x, const float* dy, const float* w, float* dx, float* dw, float* db, int B,
                             int C, int P, int kind, int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && dy && w && dx && dw && B > 0 && C > 0 && P > 0 && (kind == 1 || kind == 2), JIRF_EINVAL,
                 "layernorm_bwd: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = (P + 255) / 256;
    const int slots = gx * B;
    DeviceBuffer pw, pb;
    pw.ensure((size_t)slots * C * sizeof(double));
    pb.ensure((size_t)slots * C * sizeof(double));
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(gx, B), dim3(256), 256 * sizeof(double), s, x, (long)C * P, dy,
                       (long)C * P, w, dx, (long)C * P, static_cast<double*>(pw.p), static_cast<double*>(pb.p), C, P,
                       kind == 1 ? 1 : 0);
    hipLaunchKernelGGL(reduce_parts_d_kernel, dim3((C + 255) / 256), dim3(256), 0, s, static_cast<const double*>(pw.p),
                       (long)C, slots, dw, accumulate);
    if (db && kind == 1)
      hipLaunchKernelGGL(reduce_parts_d_kernel, dim3((C + 255) / 256), dim3(256), 0, s,
                         static_cast<const double*>(pb.p), (long)C, slots, db, accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_train_gelu_bwd(const float* pre, const float* x2, const float* dy, float* dx, int B, int C, int P, int mode,
                        void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(pre && dy && dx && B > 0 && C > 0 && P > 0 && (mode == 0 || (mode == 1 && x2)), JIRF_EINVAL,
                 "gelu_bwd: bad arguments");
    const long n_img = (long)C * P;
    const int gx = (int)std::min<long>(2048, (n_img + 255) / 256);
    // mode 1: pre / x2 are the two halves [C][P] of one [2C][P] tensor per image, dx likewise
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), pre, x2, dy, dx,
                       n_img, mode ? 2 * n_img : n_img, mode ? 2 * n_img : 0L, n_img, mode ? 2 * n_img : n_img, n_img, mode);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_dwconv3x3_bwd_weight(const float* x, const float* dy, float* dw, float* db, int B, int C, int H, int W,
                                    int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && dy && dw && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "dwconv3x3_bwd_weight: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceBuffer part;
    part.ensure((size_t)B * C * 10 * sizeof(double));
    hipLaunchKernelGGL(dwconv_dw_kernel, dim3(C, B), dim3(256), 0, s, x, (long)C * H * W, dy, (long)C * H * W,
                       static_cast<double*>(part.p), C, H, W);
    hipLaunchKernelGGL(dwconv_dw_reduce_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, s,
                       static_cast<const double*>(part.p), B, C, dw, db, accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

}  // extern "C"
