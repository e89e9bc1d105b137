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
#include <atomic>
#include <cmath>
#include <memory>
#include <mutex>
#include <vector>

#include "runtime.h"

namespace jirf {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_f(float v) { return gelu_fast(v); }
// d/dv [0.5 v (1 + erf(v / sqrt 2))] = 0.5 (1 + erf(v / sqrt 2)) + v exp(-v^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_grad_f(float v) { return gelu_grad_fast(v); }

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
// out = scale * sum(part[0..n)): one wave, lane l adds parts l, l + 64, ..., then a butterfly of fixed shape.
__global__ __launch_bounds__(64) void sum_parts_kernel(const double* __restrict__ part, int n, double* __restrict__ out,
                                                       double scale, double* __restrict__ out2) {
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) t += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (threadIdx.x == 0) {
    out[0] = t * scale;
    if (out2) out2[0] = t * scale;
  }
}

// Gradient clip + Adam on flat buffers, Jittor's update (jt.optim.Adam.step):
//   g <- grad_scale * g                                            (the 1 / ranks of the data-parallel mean; 1: off)
//   g <- g * min(max_norm / (||g|| + 1e-6), 1)                     (Optimizer.clip_grad_norm; max_norm <= 0: off)
//   m <- b0 m + (1 - b0) g ;  v <- b1 v + (1 - b1) g^2
//   p <- p - m * lr * sqrt(1 - b1^t) / (1 - b0^t) / (sqrt(v) + eps)      (eps outside the bias correction)
// torch_form != 0 selects torch's placement instead: p <- p - lr / (1 - b0^t) * m / (sqrt(v) / sqrt(1 - b1^t) + eps).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const double* __restrict__ sumsq, float max_norm, float grad_scale,
                                                   float lr, float b0, float b1, float eps, float bc0, float bc1,
                                                   int torch_form) {
  // sumsq is the sum of squares of the buffer as it stands, i.e. before grad_scale: ||grad_scale g|| = grad_scale ||g||
  float clip = grad_scale;
  if (max_norm > 0.f && sumsq) clip *= fminf(max_norm / (grad_scale * (float)sqrt(sumsq[0]) + 1e-6f), 1.0f);
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
// CONV: the same contraction for a k x k convolution, dW[o][c][ky][kx] = sum dY[b][o][oy][ox] X[b][c][oy s - pad + ky]
// [ox s - pad + kx]: row k = c * ks^2 + ky * ks + kx of the "X" operand is gathered from the input plane
// (zero outside it), so the result comes out in the reference's [Cout][Cin][ks][ks] layout.
constexpr int NT_T = 64, NT_PX = 16, NT_LD = NT_PX + 1;
constexpr int NT_LPR = NT_PX / 4, NT_RPI = 64 / NT_LPR, NT_U = NT_T / NT_RPI;   // loader: lanes per row, rows per instruction

struct ConvGeom {
  int ks, stride, pad, Hin, Win, Wout;
};

template <bool CONV>
__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const float* __restrict__ dy, long dy_bs,
                                                           const float* __restrict__ x, long x_bs,
                                                           float* __restrict__ part, double* __restrict__ dbpart,
                                                           int B, int M, int K, int P, int chunks_per_slab,
                                                           ConvGeom cg) {
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
  // the chunk's global loads: issued one chunk ahead, so that they are in flight during the MFMA phase
  float4 va[NT_U], vb[NT_U];
  int tap_off[NT_U], tap_ky[NT_U], tap_kx[NT_U];   // CONV: the gathered rows' input plane and tap (loop-invariant)
  if constexpr (CONV) {
    const int taps = cg.ks * cg.ks;
#pragma unroll
    for (int u = 0; u < NT_U; ++u) {
      const int k = min(k0 + lr + NT_RPI * u, K - 1);
      const int ch = k / taps, t = k - ch * taps;
      tap_ky[u] = t / cg.ks;
      tap_kx[u] = t - tap_ky[u] * cg.ks;
      tap_off[u] = ch * cg.Hin * cg.Win;
    }
  }
  auto fetch = [&](long c) {
    const int b = (int)(c / chunks_img);
    const int p0 = (int)(c - (long)b * chunks_img) * NT_PX;
    const float* __restrict__ dyb = dy + (long)b * dy_bs;
    const float* __restrict__ xb = x + (long)b * x_bs;
    const int pa = min(p0 + lc, P - 4);   // clamped address; the tail is zeroed when the chunk is staged
    int gy[4], gx[4];                     // CONV: the four pixels' input coordinates at tap (0, 0)
    if constexpr (CONV) {
      const int oy0 = pa / cg.Wout, ox0 = pa - oy0 * cg.Wout;   // one division per chunk; Wout >= 4
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool wrap = ox0 + j >= cg.Wout;
        gy[j] = (oy0 + (wrap ? 1 : 0)) * cg.stride - cg.pad;
        gx[j] = (ox0 + j - (wrap ? cg.Wout : 0)) * cg.stride - cg.pad;
      }
    }
#pragma unroll
    for (int u = 0; u < NT_U; ++u) {
      const int row = lr + NT_RPI * u;
      va[u] = *reinterpret_cast<const float4*>(dyb + (long)min(m0 + row, M - 1) * P + pa);
      if constexpr (!CONV) {
        vb[u] = *reinterpret_cast<const float4*>(xb + (long)min(k0 + row, K - 1) * P + pa);
      } else {
        const float* __restrict__ plane = xb + tap_off[u];
        float g4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int sy = gy[j] + tap_ky[u], sx = gx[j] + tap_kx[u];
          const bool in = (unsigned)sy < (unsigned)cg.Hin && (unsigned)sx < (unsigned)cg.Win;
          const float v = plane[in ? sy * cg.Win + sx : 0];
          g4[j] = in ? v : 0.f;
        }
        vb[u] = make_float4(g4[0], g4[1], g4[2], g4[3]);
      }
    }
  };
  if (c_lo + wave < c_hi) fetch(c_lo + wave);
  for (long c = c_lo + wave; c < c_hi; c += 4) {
    const int p0 = (int)(c - (c / chunks_img) * chunks_img) * NT_PX;
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
    if (c + 4 < c_hi) fetch(c + 4);
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

// The same contraction with the operand chunks shared by the workgroup's four waves.  The kernel above loads
// 2 x 64 rows per 64 x 64 tile and wave (16 flop per byte moved from the caches, which is what bounds it); here a
// workgroup owns a (WM TI 32) x (WK TJ 32) tile of dW, stages the [rows][16 px] chunks of both operands once in LDS
// (double-buffered, one barrier per chunk, the next chunk's global loads in flight during the MFMA phase) and wave
// (wm, wk) accumulates its (TI 32) x (TJ 32) part: 32 flop per byte at 128 x 128.  Rows past M / K are loaded from
// the last valid row (their products land in outputs that are never written), so the loop has no row masks; a
// ragged last chunk of the plane is zero-filled.  Each output element belongs to one wave: the slab's tile is
// written from the accumulators directly; db (wk == 0, k-tile 0) likewise.
template <bool CONV, int WM, int WK, int TI, int TJ>
__global__ __launch_bounds__(256, 2) void dw_shared_kernel(const float* __restrict__ dy, long dy_bs,
                                                           const float* __restrict__ x, long x_bs,
                                                           float* __restrict__ part, double* __restrict__ dbpart,
                                                           int B, int M, int K, int P, int chunks_per_slab,
                                                           ConvGeom cg) {
  static_assert(WM * WK == 4, "four waves");
  constexpr int TM = WM * TI * 32, TK = WK * TJ * 32;
  constexpr int UA = (TM + 63) / 64, UB = (TK + 63) / 64;   // float4 per thread and chunk (64 rows per instruction)
  __shared__ float lds[2][(TM + TK) * NT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WK, wk = wave % WK;
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TK, slab = blockIdx.z;
  const int chunks_img = (P + NT_PX - 1) / NT_PX;
  const long chunks = (long)chunks_img * B;
  const long c_lo = (long)slab * chunks_per_slab, c_hi = min(chunks, c_lo + chunks_per_slab);
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  double dbacc = 0.0;   // wk == 0: row m0 + wm TI 32 + lane
  const int lr = tid >> 2, lc = (tid & 3) * 4;
  float4 va[UA], vb[UB];
  int tap_off[UB], tap_ky[UB], tap_kx[UB];   // CONV: the gathered rows' input plane and tap (loop-invariant)
  if constexpr (CONV) {
    const int taps = cg.ks * cg.ks;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int k = min(k0 + lr + 64 * u, K - 1);
      const int ch = k / taps, t = k - ch * taps;
      tap_ky[u] = t / cg.ks;
      tap_kx[u] = t - tap_ky[u] * cg.ks;
      tap_off[u] = ch * cg.Hin * cg.Win;
    }
  }
  auto fetch = [&](long c) {
    const int b = (int)(c / chunks_img);
    const int p0 = (int)(c - (long)b * chunks_img) * NT_PX;
    const float* __restrict__ dyb = dy + (long)b * dy_bs;
    const float* __restrict__ xb = x + (long)b * x_bs;
    const int pa = min(p0 + lc, P - 4);
    int gy[4], gx[4];                     // CONV: the four pixels' input coordinates at tap (0, 0)
    if constexpr (CONV) {
      const int oy0 = pa / cg.Wout, ox0 = pa - oy0 * cg.Wout;   // one division per chunk; Wout >= 4
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool wrap = ox0 + j >= cg.Wout;
        gy[j] = (oy0 + (wrap ? 1 : 0)) * cg.stride - cg.pad;
        gx[j] = (ox0 + j - (wrap ? cg.Wout : 0)) * cg.stride - cg.pad;
      }
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int row = lr + 64 * u;
      if (TM % 64 == 0 || row < TM) va[u] = *reinterpret_cast<const float4*>(dyb + (long)min(m0 + row, M - 1) * P + pa);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int row = lr + 64 * u;
      if (TK % 64 == 0 || row < TK) {
        if constexpr (!CONV) {
          vb[u] = *reinterpret_cast<const float4*>(xb + (long)min(k0 + row, K - 1) * P + pa);
        } else {
          const float* __restrict__ plane = xb + tap_off[u];
          float g4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sy = gy[j] + tap_ky[u], sx = gx[j] + tap_kx[u];
            const bool in = (unsigned)sy < (unsigned)cg.Hin && (unsigned)sx < (unsigned)cg.Win;
            const float v = plane[in ? sy * cg.Win + sx : 0];
            g4[j] = in ? v : 0.f;
          }
          vb[u] = make_float4(g4[0], g4[1], g4[2], g4[3]);
        }
      }
    }
  };
  if (c_lo < c_hi) fetch(c_lo);
  for (long c = c_lo; c < c_hi; ++c) {
    float* __restrict__ A = lds[(c - c_lo) & 1];
    float* __restrict__ Bm = A + TM * NT_LD;
    const int p0 = (int)(c - (c / chunks_img) * chunks_img) * NT_PX;
    const bool pin = p0 + lc < P;   // P % 4 == 0: a float4 is inside the plane or outside it as a whole
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int row = lr + 64 * u;
      if (TM % 64 == 0 || row < TM) {
        float* d = A + row * NT_LD + lc;
        d[0] = pin ? va[u].x : 0.f, d[1] = pin ? va[u].y : 0.f, d[2] = pin ? va[u].z : 0.f, d[3] = pin ? va[u].w : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int row = lr + 64 * u;
      if (TK % 64 == 0 || row < TK) {
        float* d = Bm + row * NT_LD + lc;
        d[0] = pin ? vb[u].x : 0.f, d[1] = pin ? vb[u].y : 0.f, d[2] = pin ? vb[u].z : 0.f, d[3] = pin ? vb[u].w : 0.f;
      }
    }
    if (c + 1 < c_hi) fetch(c + 1);
    // one barrier per chunk: the other buffer was last read in the previous trip's MFMA phase, which every wave
    // has left before it arrives here, so the next trip may overwrite it
    __syncthreads();
    if (wk == 0 && blockIdx.y == 0 && lane < TI * 32) {
      const float* __restrict__ ar = A + (wm * TI * 32 + lane) * NT_LD;
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < NT_PX; ++j) t += ar[j];
      dbacc += t;
    }
    const float* __restrict__ Aw = A + (wm * TI * 32 + l31) * NT_LD + half;
    const float* __restrict__ Bw = Bm + (wk * TJ * 32 + l31) * NT_LD + half;
#pragma unroll
    for (int q = 0; q < NT_PX / 2; ++q) {
      float af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = Aw[i * 32 * NT_LD + 2 * q];
#pragma unroll
      for (int j = 0; j < TJ; ++j) bf[j] = Bw[j * 32 * NT_LD + 2 * q];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  float* __restrict__ out = part + (long)slab * M * K;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = k0 + (wk * TJ + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M && col < K) out[(long)row * K + col] = acc[i][j][r];
      }
    }
  if (wk == 0 && blockIdx.y == 0 && dbpart && lane < TI * 32) {
    const int row = m0 + wm * TI * 32 + lane;
    if (row < M) dbpart[(long)slab * M + row] = dbacc;
  }
}

// out[e] (+)= sum over slabs of part[slab][e], and in the same launch db[c] (+)= sum over slabs of dbpart[slab][c]
// (workgroups w_blocks and up).  vec: a thread owns four consecutive outputs and adds the slabs in order (large n,
// few slabs: coalesced 16-byte reads).  Otherwise a workgroup owns 16 consecutive outputs; its 16 thread groups add
// every 16th slab each and the 16 sums meet in LDS in group order.  Either way the order is fixed: reproducible.
__global__ __launch_bounds__(256) void reduce_dw_kernel(const float* __restrict__ part, long n, int slabs,
                                                        float* __restrict__ out, const double* __restrict__ dbpart,
                                                        int M, float* __restrict__ db, unsigned w_blocks, int vec,
                                                        int accumulate) {
  __shared__ double red[16][16];
  const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
  if (blockIdx.x >= w_blocks) {
    const int c = (int)(blockIdx.x - w_blocks) * 16 + col;
    double t = 0.0;
    if (c < M)
      for (int s = grp; s < slabs; s += 16) t += dbpart[(long)s * M + c];
    red[grp][col] = t;
    __syncthreads();
    if (grp == 0 && c < M) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += red[g][col];
      db[c] = accumulate ? db[c] + (float)v : (float)v;
    }
    return;
  }
  if (vec) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < slabs; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long)s * n + e);
      t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(out + e);
    if (accumulate) {
      const float4 v = *o;
      t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
    }
    *o = t;
    return;
  }
  float* redf = reinterpret_cast<float*>(&red[0][0]);
  const long e = (long)blockIdx.x * 16 + col;
  float t = 0.f;
  if (e < n)
    for (int s = grp; s < slabs; s += 16) t += part[(long)s * n + e];
  redf[grp * 16 + col] = t;
  __syncthreads();
  if (grp == 0 && e < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += redf[g * 16 + col];
    out[e] = accumulate ? out[e] + v : v;
  }
}
// out[c] (+)= sum over slots of part[slot][c], one workgroup per column: thread t adds slots t, t + 256, ... and
// the 256 sums meet in a tree of fixed shape, so the result does not depend on timing.  blockIdx.y = 1 does the
// same for a second (part, out) pair of the same shape.
__global__ __launch_bounds__(256) void reduce_cols_d_kernel(const double* __restrict__ part0, float* __restrict__ out0,
                                                            const double* __restrict__ part1, float* __restrict__ out1,
                                                            int n, int slots, int accumulate) {
  __shared__ double red[256];
  const double* __restrict__ part = blockIdx.y ? part1 : part0;
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  const int c = blockIdx.x;
  double t = 0.0;
  for (int s = threadIdx.x; s < slots; s += 256) t += part[(long)s * n + c];
  t = block_sum_d(t, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)t : (float)t;
}

// ---- LayerNorm over channels, backward ----------------------------------------------------------------
// n = (x - mu) rs (WithBias) or x rs (BiasFree, hint/astv2; AST's nn.LayerNorm is the WithBias form);
// y = n w + b.  With g = dy w per channel:
//   WithBias:  dx = rs (g - mean_c(g) - nhat mean_c(g nhat)),  nhat = (x - mu) rs
//   BiasFree:  dx = rs g - (x - mu) rs^3 mean_c(g x)          (the variance still centres x)
// A workgroup of NW waves owns 64 pixels (lane = pixel); wave w takes the channels w, w + NW, ... in both passes,
// four of them per loop trip so that eight loads per lane are in flight.  Pass 1 sums x, x^2, g and g x about the
// pixel's first channel (shifted data keeps the one-pass variance as accurate as the two-pass form); the waves'
// sums meet in LDS and are added in wave order.  Pass 2 writes dx and reduces the parameter gradients
// dw[c] = sum dy n, db[c] = sum dy over the 64 pixels with cross-lane shuffles: one partial per (workgroup,
// channel), added up by reduce_cols_d_kernel.  NW = 16 where the plane is small (few
// workgroups, many channels each), 4 otherwise.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_kernel(const float* __restrict__ x, long x_bs,
                                                                const float* __restrict__ dy, long dy_bs,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ add, long add_bs,
                                                                float* __restrict__ dx, long dx_bs,
                                                                double* __restrict__ part_w,
                                                                double* __restrict__ part_b, int C, int P,
                                                                int with_bias) {
  __shared__ float st[NW][4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const bool on = p < P;
  const long off = on ? p : 0;
  const float* __restrict__ xb = x + (long)b * x_bs + off;
  const float* __restrict__ gb = dy + (long)b * dy_bs + off;
  const float piv = xb[0];
  float sx = 0.f, sxx = 0.f, sg = 0.f, sgx = 0.f;
  for (int c0 = wave; c0 < C; c0 += 4 * NW) {
    float xv[4], gv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(c0 + j * NW, C - 1);
      xv[j] = xb[(long)c * P];
      gv[j] = gb[(long)c * P] * w[c];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j * NW < C) {
        const float d = xv[j] - piv;
        sx += d;
        sxx = fmaf(d, d, sxx);
        sg += gv[j];
        sgx = fmaf(gv[j], d, sgx);
      }
  }
  st[wave][0][lane] = sx;
  st[wave][1][lane] = sxx;
  st[wave][2][lane] = sg;
  st[wave][3][lane] = sgx;
  __syncthreads();
  sx = sxx = sg = sgx = 0.f;
#pragma unroll
  for (int v = 0; v < NW; ++v) {
    sx += st[v][0][lane];
    sxx += st[v][1][lane];
    sg += st[v][2][lane];
    sgx += st[v][3][lane];
  }
  const float invC = 1.0f / (float)C;
  const float md = sx * invC;                       // mean of (x - pivot)
  const float var = fmaxf(sxx * invC - md * md, 0.f);
  const float mu = piv + md;
  const float rs = 1.0f / sqrtf(var + 1e-5f);
  const float s1 = sg * invC;
  // WithBias: mean_c(g nhat) = rs mean_c(g (x - mu));  BiasFree: mean_c(g x)
  const float s2 = with_bias ? rs * (sgx - md * sg) * invC : (sgx + piv * sg) * invC;
  float* __restrict__ dxb = dx + (long)b * dx_bs + off;
  const float* __restrict__ adb = add ? add + (long)b * add_bs + off : nullptr;   // + gradient of the shortcut
  const long slot = ((long)blockIdx.y * gridDim.x + blockIdx.x) * C;
  for (int c0 = wave; c0 < C; c0 += 4 * NW) {
    float xv[4], dv[4], av[4], wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(c0 + j * NW, C - 1);
      xv[j] = xb[(long)c * P];
      dv[j] = gb[(long)c * P];
      av[j] = adb ? adb[(long)c * P] : 0.f;
      wv[j] = w[c];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + j * NW;
      if (c >= C) break;               // uniform over the wave
      const float g = dv[j] * wv[j];
      const float nh = (xv[j] - mu) * rs;
      float o;
      if (with_bias) o = rs * (g - s1 - nh * s2);
      else o = rs * g - (xv[j] - mu) * rs * rs * rs * s2;
      if (on) dxb[(long)c * P] = o + av[j];
      const float tw = wave_sum_f(on ? dv[j] * (with_bias ? nh : xv[j] * rs) : 0.f);
      const float tb = wave_sum_f(on ? dv[j] : 0.f);
      if (lane == 0) {
        part_w[slot + c] = (double)tw;
        part_b[slot + c] = (double)tb;
      }
    }
  }
}

// ---- elementwise pieces of the token MLPs -------------------------------------------------------------------
// One image = n_img contiguous floats at pointer + b * batch stride (a channel slice of a wider tensor is a
// pointer offset plus that tensor's batch stride).
// forward   mode 0: y = gelu(pre)                    (after linear1, ast_arch.py:384,456; after the depthwise 3x3 of LeFF)
//           mode 1: y = gelu(pre) * x2               (the FRFN gate, ast_arch.py:507-510)
//           mode 3: y = gelu(pre) * gelu(x2)         (the same gate fed linear1's raw second half: the GELU after
//                                                     linear1 is then applied only where the depthwise 3x3 needs it)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ pre, long pre_bs,
                                                       const float* __restrict__ x2, long x2_bs, float* __restrict__ y,
                                                       long y_bs, long n_img, int mode) {
  const int b = blockIdx.y;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_img; i += (long)gridDim.x * 256) {
    const float a = gelu_f(pre[(long)b * pre_bs + i]);
    if (mode == 0) y[(long)b * y_bs + i] = a;
    else if (mode == 1) y[(long)b * y_bs + i] = a * x2[(long)b * x2_bs + i];
    else y[(long)b * y_bs + i] = a * gelu_f(x2[(long)b * x2_bs + i]);
  }
}
// the same, four adjacent elements per thread as one 16-byte access per tensor (n_img % 4 == 0, 16-byte aligned rows)
__global__ __launch_bounds__(256) void gelu_fwd4_kernel(const float* __restrict__ pre, long pre_bs,
                                                        const float* __restrict__ x2, long x2_bs, float* __restrict__ y,
                                                        long y_bs, long n4, int mode) {
  const int b = blockIdx.y;
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(pre + (long)b * pre_bs);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x2 ? x2 + (long)b * x2_bs : pre);
  float4* __restrict__ y4 = reinterpret_cast<float4*>(y + (long)b * y_bs);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 a = p4[i];
    float4 r = make_float4(gelu_f(a.x), gelu_f(a.y), gelu_f(a.z), gelu_f(a.w));
    if (mode != 0) {
      float4 g = x4[i];
      if (mode == 3) g = make_float4(gelu_f(g.x), gelu_f(g.y), gelu_f(g.z), gelu_f(g.w));
      r = make_float4(r.x * g.x, r.y * g.y, r.z * g.z, r.w * g.w);
    }
    y4[i] = r;
  }
}
// backward  mode 0: dxa = dy * gelu'(pre)
//           mode 1: y = gelu(a) * b with a = pre, b = x2:  dxa = dy * b * gelu'(a),  dxb = dy * gelu(a)
//           mode 3: y = gelu(a) * gelu(b) with a = pre, b = x2 (both raw):  dxa = dy * gelu(b) * gelu'(a),
//                   dxb = dy * gelu(a) * gelu'(b)
//           mode 2: LeakyReLU(0.01) (ast_arch.py:563 input_proj), pre = the activation's OUTPUT (same sign as its
//                   input): dxa = dy * (pre > 0 ? 1 : 0.01)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ pre, long pre_bs,
                                                       const float* __restrict__ x2, long x2_bs,
                                                       const float* __restrict__ dy, long dy_bs, float* __restrict__ dxa,
                                                       long dxa_bs, float* __restrict__ dxb, long dxb_bs, long n_img,
                                                       int mode) {
  const int b = blockIdx.y;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_img; i += (long)gridDim.x * 256) {
    const float a = pre[(long)b * pre_bs + i], g = dy[(long)b * dy_bs + i];
    if (mode == 0) {
      dxa[(long)b * dxa_bs + i] = g * gelu_grad_f(a);
    } else if (mode == 1) {
      const float bv = x2[(long)b * x2_bs + i];
      dxa[(long)b * dxa_bs + i] = g * bv * gelu_grad_f(a);
      dxb[(long)b * dxb_bs + i] = g * gelu_f(a);
    } else if (mode == 3) {
      const float br = x2[(long)b * x2_bs + i];
      dxa[(long)b * dxa_bs + i] = g * gelu_f(br) * gelu_grad_f(a);
      dxb[(long)b * dxb_bs + i] = g * gelu_f(a) * gelu_grad_f(br);
    } else {
      dxa[(long)b * dxa_bs + i] = a > 0.f ? g : 0.01f * g;
    }
  }
}
// y = a (+ b): the copies of the concatenations and the sums of gradients that meet at a shortcut.
// scale != null: y = a + scale[image] * b (a may be null): a branch under stochastic depth, whose output (and
// output gradient) is multiplied per sample by 0 or 1 / keep_prob (DropPath, ast_arch.py:52-70).
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, long a_bs, const float* __restrict__ b2,
                                                  long b_bs, const float* __restrict__ scale, float* __restrict__ y,
                                                  long y_bs, long n_img) {
  const int b = blockIdx.y;
  const float sc = scale ? scale[b] : 1.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_img; i += (long)gridDim.x * 256) {
    float v = a ? a[(long)b * a_bs + i] : 0.f;
    if (b2) v = fmaf(sc, b2[(long)b * b_bs + i], v);
    y[(long)b * y_bs + i] = v;
  }
}
// Stochastic depth: out[row][b] = 0 with probability rate[row], else 1 / (1 - rate[row]) (drop_path(), reference
// ast_arch.py:62-70).  Counter-based: the uniform variate of (row, b) at draw number counter[0] is a hash of
// (seed, counter, row, b), so a sequence depends on the seed alone, not on how the launches are issued.  One
// workgroup; it advances the counter when every thread has read it.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void drop_scales_kernel(float* __restrict__ out, const float* __restrict__ rate, int rows,
                                                          int B, unsigned long long seed, unsigned long long* counter) {
  const unsigned long long draw = counter[0];
  __syncthreads();
  for (int e = threadIdx.x; e < rows * B; e += 256) {
    const int row = e / B;
    const unsigned long long h = mix64(mix64(seed ^ (draw * 0xD1342543DE82EF95ull)) + (unsigned long long)e);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);   // 24 bits: [0, 1)
    const float keep = 1.0f - rate[row];
    out[e] = u < keep ? 1.0f / keep : 0.f;
  }
  if (threadIdx.x == 0) counter[0] = draw + 1;
}
// out[c] (+)= sum_{b,p} x[b][c][p] (the bias gradient of the transposed convolution): one workgroup per channel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out,
                                                          int B, long P, int accumulate) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* __restrict__ xp = x + (long)b * x_bs + (long)c * P;
    for (long i = threadIdx.x; i < P; i += 256) acc += xp[i];
  }
  const double t = block_sum_d(acc, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)t : (float)t;
}
// w [C][9] -> taps reversed (the depthwise data gradient is the forward kernel on these)
__global__ void flip9_kernel(const float* __restrict__ w, float* __restrict__ out, int C) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < C * 9) out[e] = w[(e / 9) * 9 + 8 - e % 9];
}

// ---- depthwise 3x3, weight and bias gradients -------------------------------------------------------------
// dw[c][ky][kx] = sum_{b,y,x} dy[b][c][y][x] * x[b][c][y + ky - 1][x + kx - 1] (zero padded), db[c] = sum dy.
// One workgroup per (channel, image, slab of rows): a thread owns four adjacent pixels of a row (float4 of dy,
// float4 of the three input rows plus their two outer neighbours: 9 + 2 loads per four pixels instead of ten per
// pixel), accumulates its 9 + 1 sums in fp32 (a few dozen terms each) and the workgroup adds them in double;
// part [B * slabs][C][10], reduced in (image, slab) order.  W % 4 == 0 takes the vector path.
__global__ __launch_bounds__(256) void dwconv_dw_kernel(const float* __restrict__ x, long x_bs,
                                                        const float* __restrict__ dy, long dy_bs,
                                                        double* __restrict__ part, int C, int H, int W, int slabs) {
  __shared__ double red[256];
  const int c = blockIdx.x, b = blockIdx.y / slabs, slab = blockIdx.y - b * slabs;
  const float* __restrict__ xp = x + (long)b * x_bs + (long)c * H * W;
  const float* __restrict__ gp = dy + (long)b * dy_bs + (long)c * H * W;
  float acc[10];
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  const int rows_per = (H + slabs - 1) / slabs;
  const int y_lo = slab * rows_per, y_hi = min(H, y_lo + rows_per);
  const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(gp)) & 15) == 0;
  // rows of 4 .. 64 float4 that are a power of two: a row's float4s are consecutive lanes of one wave, so the two outer
  // neighbours of a lane's four pixels are its neighbour lanes' edge pixels (whole-wave DPP shifts, zero at the row
  // ends) and the position needs no division
  const int w4 = W >> 2;
  const bool pow2 = vec && w4 >= 1 && w4 <= 64 && (w4 & (w4 - 1)) == 0;
  if (pow2) {
    const int sh = 31 - __builtin_clz(w4), n4 = (y_hi - y_lo) * w4;
    const int n4r = (n4 + 255) & ~255;                      // every lane of a wave takes part in the shifts
    for (int i = threadIdx.x; i < n4r; i += 256) {
      const bool on = i < n4;
      const int ii = on ? i : 0;
      const int yy = y_lo + (ii >> sh), x0 = (ii & (w4 - 1)) * 4;
      float4 g = *reinterpret_cast<const float4*>(gp + (long)yy * W + x0);
      if (!on) g = make_float4(0.f, 0.f, 0.f, 0.f);
      const float gv[4] = {g.x, g.y, g.z, g.w};
      acc[9] += (g.x + g.y) + (g.z + g.w);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int sy = yy + ky - 1;
        const bool rin = sy >= 0 && sy < H;
        const float4 v0 = *reinterpret_cast<const float4*>(xp + (long)min(max(sy, 0), H - 1) * W + x0);
        const float4 v = rin ? v0 : make_float4(0.f, 0.f, 0.f, 0.f);
        float xl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0x138, 0xf, 0xf, true));
        float xr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0x130, 0xf, 0xf, true));
        if (x0 == 0) xl = 0.f;
        if (x0 + 4 >= W) xr = 0.f;
        const float xv[6] = {xl, v.x, v.y, v.z, v.w, xr};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = fmaf(gv[j], xv[j + kx], acc[ky * 3 + kx]);
      }
    }
  } else if (vec) {
    const int n4 = (y_hi - y_lo) * w4;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const int yy = y_lo + i / w4, x0 = (i % w4) * 4;
      const float4 g = *reinterpret_cast<const float4*>(gp + (long)yy * W + x0);
      const float gv[4] = {g.x, g.y, g.z, g.w};
      acc[9] += (g.x + g.y) + (g.z + g.w);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int sy = yy + ky - 1;
        if (sy < 0 || sy >= H) continue;
        const float* __restrict__ row = xp + (long)sy * W + x0;
        const float4 v = *reinterpret_cast<const float4*>(row);
        const float xl = x0 > 0 ? row[-1] : 0.f, xr = x0 + 4 < W ? row[4] : 0.f;
        const float xv[6] = {xl, v.x, v.y, v.z, v.w, xr};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = fmaf(gv[j], xv[j + kx], acc[ky * 3 + kx]);
      }
    }
  } else {
    for (int i = y_lo * W + threadIdx.x; i < y_hi * W; i += 256) {
      const int yy = i / W, xx = i - yy * W;
      const float g = gp[i];
      acc[9] += g;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int sy = yy + ky - 1, sx = xx + kx - 1;
          if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc[ky * 3 + kx] = fmaf(g, xp[sy * W + sx], acc[ky * 3 + kx]);
        }
    }
  }
  for (int t = 0; t < 10; ++t) {
    const double v = block_sum_d((double)acc[t], red);
    if (threadIdx.x == 0) part[((long)blockIdx.y * C + c) * 10 + t] = v;
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


// ---- weights of the training GEMMs, packed on the device -----------------------------------------------------
// The forward GEMM family reads K-major packed weights wt[k][o] (k rows padded to 16, row length to 128).
// In training the weights live in a flat device buffer in the reference's layouts and change every step, so
// every call packs what it needs: the forward operand, or the operand of the data gradient.
//   kind 0  forward convolution           w [M][Cin][taps]:            wt[tap Cin + c][o] = w[o][c][tap]
//   kind 1  data gradient, stride 1       w [Cin][M][taps] (forward):  wt[tap Cin + c][o] = w[c][o][taps - 1 - tap]
//   kind 2  ConvTranspose2d(2, stride 2)  w [Cin][M / 4][2][2]:        wt[c][o] = w[c][o]            (o = 4 co + tap)
//   kind 4  data gradient of Conv2d(4, stride 2, pad 1), forward w [Cin][M / 4][4][4]: as a 3 x 3 convolution over
//           the gradient plane whose row o = 4 ci + 2 a + b is output pixel (2 i + a, 2 j + b):
//           wt[tap3 Cin + c][o] = w[c][ci][a + 1 - 2 dy][b + 1 - 2 dx] where that tap exists (tap3 = 3 (dy + 1) + dx + 1)
__global__ __launch_bounds__(256) void pack_train_kernel(const float* __restrict__ w, float* __restrict__ wt, int kind,
                                                         int M, int Cin, int taps, int K, int Kp, int Mp) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)Kp * Mp) return;
  const int k = (int)(e / Mp), o = (int)(e - (long)k * Mp);
  float v = 0.f;
  if (k < K && o < M) {
    const int tap = k / Cin, c = k - tap * Cin;
    if (kind == 0) v = w[((long)o * Cin + c) * taps + tap];
    else if (kind == 1) v = w[((long)c * M + o) * taps + (taps - 1 - tap)];
    else if (kind == 2) v = w[(long)k * M + o];
    else {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int ci = o >> 2, ky = ((o >> 1) & 1) + 1 - 2 * dy, kx = (o & 1) + 1 - 2 * dx;
      if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = w[(((long)c * (M >> 2) + ci) * 4 + ky) * 4 + kx];
    }
  }
  wt[e] = v;
}
__global__ void expand_bias4_kernel(const float* __restrict__ b, float* __restrict__ out, int M) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < M) out[e] = b[e >> 2];
}

// All the packed forms of a step in one launch.  An item is one (weight, kind) pair; its packed matrix is cut into
// 64 x 64 tiles of wt[k][o], one workgroup each (block0 = the item's first workgroup; the item's last workgroup
// also expands the bias of a transposed convolution behind the matrix).  The tile goes through LDS so that both
// sides are read and written in runs: the source is walked with its contiguous index fastest (kind 0: k, the
// others: o), the packed matrix with o fastest.
struct PackItem {
  const float* w;
  const float* bias;   // kind 2 only (or null): expanded to [4 Cout] at wt + Kp * Mp
  float* wt;
  int kind, M, Cin, taps, K, Kp, Mp;
  unsigned block0, tiles_o;
};
__device__ __forceinline__ float pack_source(const PackItem& it, int k, int o) {
  if (k >= it.K || o >= it.M) return 0.f;
  const int tap = k / it.Cin, c = k - tap * it.Cin;
  if (it.kind == 0) return it.w[((long)o * it.Cin + c) * it.taps + tap];
  if (it.kind == 1) return it.w[((long)c * it.M + o) * it.taps + (it.taps - 1 - tap)];
  if (it.kind == 2) return it.w[(long)k * it.M + o];
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int ci = o >> 2, ky = ((o >> 1) & 1) + 1 - 2 * dy, kx = (o & 1) + 1 - 2 * dx;
  return (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) ? it.w[(((long)c * (it.M >> 2) + ci) * 4 + ky) * 4 + kx] : 0.f;
}
__global__ __launch_bounds__(256) void pack_plan_kernel(const PackItem* __restrict__ items, int n_items) {
  __shared__ float tile[64][65];
  __shared__ PackItem it;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_items - 1;   // the last item whose block0 <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].block0 <= blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    it = items[lo];
  }
  __syncthreads();
  const unsigned t = blockIdx.x - it.block0;
  const int k0 = (int)(t / it.tiles_o) * 64, o0 = (int)(t % it.tiles_o) * 64;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int fast = e & 63, slow = e >> 6;
    const int kk = it.kind == 0 ? fast : slow, oo = it.kind == 0 ? slow : fast;
    tile[kk][oo] = pack_source(it, k0 + kk, o0 + oo);
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int e = threadIdx.x + 256 * i;
    const int oo = e & 63, kk = e >> 6;
    if (k0 + kk < it.Kp && o0 + oo < it.Mp) it.wt[(long)(k0 + kk) * it.Mp + o0 + oo] = tile[kk][oo];
  }
  if (it.bias && t + 1 == it.tiles_o * (unsigned)((it.Kp + 63) / 64))
    for (int e = threadIdx.x; e < it.M; e += 256) it.wt[(long)it.Kp * it.Mp + e] = it.bias[e >> 2];
}

// ---- sparse window attention, backward (ast_arch.py:275-318; forward kernel: win_attn.hip) ------------------
// One wave per workgroup; workgroup (chunk, head) walks over the windows chunk, chunk + chunks, ... of its
// head.  For one (window, head), with n = query token, m = key token of the 8 x 8 window:
//   S[n][m] = q_n k_m + bias + mask,  P0 = softmax_m S,  P1 = relu(S)^2,  A = w0 P0 + w1 P1,  O[n] = sum_m A[n][m] v_m
//   dA[n][m] = dO_n v_m;   dS = w0 P0 (dA - sum_m dA P0) + 2 w1 relu(S) dA
//   dq_n = scale sum_m dS[n][m] k_m;   dv_m = sum_n A[n][m] dO_n;   dk_m = sum_n dS[n][m] (scale q_n)
//   d table[idx(n, m)][head] += dS[n][m];   d(w0, w1) += sum dA (P0, P1)
// All seven 64 x 64 x 32 / 64 x 32 x 64 products run on fp32 MFMA (32x32x2), following the forward kernel: the
// q, k, v, dO fragments are loaded from global memory in operand layout (lane = token, k = channel pair), and a
// lane's fragment serves as A or as B operand alike.  The score tile is formed twice:
//   "T" form  S^T[m][n] = K Q^T and dA^T = V dO^T (keys in the registers, the lane's column = one query): the
//             softmax statistics of a query (max, 1 / sum, D = sum_m dA P0) are in-lane sums plus one exchange
//             with lane ^ 32; dS^T stays in the registers and is the B operand of dQ^T = K^T dS^T (A = K rows
//             from the wave's LDS tile, lane = channel);
//   "N" form  S[n][m] = Q K^T and dA = dO V^T (queries in the registers, the lane's column = one key), with
//             the statistics read back from LDS per query: A and dS in this form are the B operands of
//             dV^T = dO^T A and dK^T = (scale Q)^T dS (A operands = dO / Q rows from LDS).
// Recomputing the tile in the other orientation costs 128 MFMAs; it replaces a 64 x 64 transpose through LDS per
// product.  k comes from the (window, head) pair the reference's flattening pairs with this one (see the forward
// kernel), and dk goes back there: the pairing is a bijection, so every gradient element has one writer.  The
// table gradient is gathered from the dS tile in LDS (lane = table entry: no two lanes ever add to one address),
// kept with the mixing-weight sums per workgroup and added up afterwards in a fixed order.
constexpr int WB_LD = 40, WB_TLD = 65;   // operand tiles: rows m and m + 4 on disjoint bank halves (as in the forward)
struct WinBwdArgs {
  int B, C, H, W, heads, shift, coupled, chunks;
  float scale;
  const float *q, *k, *v, *dout;      // planar slices, batch strides below
  long qkv_bs, dout_bs, dqkv_bs;
  float *dq, *dk, *dv;
  const float* table;                 // [225][heads]
  const float* w2;                    // [2] raw mixing weights
  double* part_table;                 // [heads][chunks][225]
  double* part_w;                     // [heads][chunks][2]
};

__global__ __launch_bounds__(64) void win_attn_bwd_kernel(const WinBwdArgs a) {
  // (the K tile is only read in the "T" form and the dS tile only written after it: they share one region,
  //  which brings a workgroup under 40 KB of LDS, four per CU)
  __shared__ float KT[64 * WB_TLD];
  __shared__ float Gs[64 * WB_LD], Qs[64 * WB_LD];
  float* __restrict__ Ks = KT;
  float* __restrict__ T = KT;
  __shared__ float ts[232];
  __shared__ float st[3][64];
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5, chunk = blockIdx.x, h = blockIdx.y;
  const int H = a.H, W = a.W, heads = a.heads;
  const int nwx = W >> 3, nW = (H >> 3) * nwx;
  const long P = (long)H * W;
  const int n_win = a.B * nW;
  const float e0 = expf(a.w2[0]), e1 = expf(a.w2[1]);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  for (int i = lane; i < 232; i += 64) ts[i] = i < 225 ? a.table[i * heads + h] : 0.f;
  float tacc[4] = {0.f, 0.f, 0.f, 0.f};   // table entries lane, lane + 64, lane + 128, lane + 192
  double g0 = 0.0, g1 = 0.0;
  auto rowmap = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * half; };   // row of accumulator register r
  for (int gw = chunk; gw < n_win; gw += a.chunks) {
    const int img = gw / nW, w = gw - img * nW;
    const int span = a.coupled ? a.B * nW : nW;
    const int j = (a.coupled ? gw : w) * heads + h;
    const int hk = j / span, bk = j - hk * span;
    const int img_k = a.coupled ? bk / nW : img;
    const int wk = a.coupled ? bk - img_k * nW : bk;
    auto pixel = [&](int win, int t) {
      const int wy = win / nwx, wx = win - wy * nwx;
      int y = wy * 8 + (t >> 3) + a.shift, x = wx * 8 + (t & 7) + a.shift;
      if (y >= H) y -= H;
      if (x >= W) x -= W;
      return (long)y * W + x;
    };
    // ---- operand fragments: token t2 * 32 + l31, channels 2 s + half -------------------------------------------
    // (addresses as a wave-uniform base + a 32-bit byte offset per lane: one register per token instead of a 64-bit
    //  pointer per load, which is what pushed this kernel's 230 loads and stores over the register file)
    auto at = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    const float* __restrict__ kb = a.k + (long)img_k * a.qkv_bs + (long)(hk * 32) * P;
    const float* __restrict__ qb = a.q + (long)img * a.qkv_bs + (long)(h * 32) * P;
    const float* __restrict__ vb = a.v + (long)img * a.qkv_bs + (long)(h * 32) * P;
    const float* __restrict__ gb = a.dout + (long)img * a.dout_bs + (long)(h * 32) * P;
    float kf[2][16], qf[2][16], vf[2][16], gf[2][16];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const unsigned oq = (unsigned)((long)half * P + pixel(w, t2 * 32 + l31)) * 4u;
      const unsigned ok = (unsigned)((long)half * P + pixel(wk, t2 * 32 + l31)) * 4u;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        kf[t2][s] = at(kb + (long)(2 * s) * P, ok);
        qf[t2][s] = at(qb + (long)(2 * s) * P, oq) * a.scale;
        vf[t2][s] = at(vb + (long)(2 * s) * P, oq);
        gf[t2][s] = at(gb + (long)(2 * s) * P, oq);
      }
    }
    __syncthreads();   // the previous window's tiles are no longer read
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int at = (t2 * 32 + l31) * WB_LD + 2 * s + half;
        Ks[at] = kf[t2][s];
        Gs[at] = gf[t2][s];
        Qs[at] = qf[t2][s];
      }
    int rid_tok = 0;   // Swin shift-mask region of token `lane` on the shifted grid
    if (a.shift) {
      const int wy = w / nwx, wx = w - wy * nwx;
      const int sy = wy * 8 + (lane >> 3), sx = wx * 8 + (lane & 7);
      const int rh = sy < H - 8 ? 0 : (sy < H - a.shift ? 1 : 2);
      const int rw = sx < W - 8 ? 0 : (sx < W - a.shift ? 1 : 2);
      rid_tok = 3 * rh + rw;
    }
    __syncthreads();   // the staged tiles; the previous window's statistics are no longer read
    // ---- "T" form: keys in the registers, the lane's column is a query; one half of the queries at a time ------
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x16 S[2], dA[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[mt][r] = 0.f; dA[mt][r] = 0.f; }
      // the query-side fragments (scaled q, dO) are read back from their staged tiles: holding them in registers
      // next to the key-side ones overflowed the register file into scratch
      const float* __restrict__ qs_frag = &Qs[(nt * 32 + l31) * WB_LD + half];
      const float* __restrict__ gs_frag = &Gs[(nt * 32 + l31) * WB_LD + half];
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          S[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[mt][s], qs_frag[2 * s], S[mt], 0, 0, 0);
          dA[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[mt][s], gs_frag[2 * s], dA[mt], 0, 0, 0);
        }
      const int n = nt * 32 + l31;
      const int ny = n >> 3, nx = n & 7;
      const int rid_n = __shfl(rid_tok, n);
      float mx = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mt * 32 + rowmap(r);
          float sc = S[mt][r] + ts[(ny - (m >> 3) + 7) * 15 + (nx - (m & 7) + 7)];
          if (a.shift) {
            const int rid_m = __shfl(rid_tok, m);
            if (rid_m != rid_n) sc += -100.0f;
          }
          S[mt][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float ex[2][16];   // each exponential once
      float sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ex[mt][r] = fast_exp(S[mt][r] - mx);
          sum += ex[mt][r];
        }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      float D = 0.f, t1 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float rl = fmaxf(S[mt][r], 0.f);
          ex[mt][r] *= inv;
          D = fmaf(dA[mt][r], ex[mt][r], D);
          t1 = fmaf(dA[mt][r], rl * rl, t1);
        }
      D += __shfl_xor(D, 32);
      if (half == 0) g0 += D;      // (both halves hold the query's full sum)
      g1 += t1;
      st[0][n] = mx;
      st[1][n] = inv;
      st[2][n] = D;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float rl = fmaxf(S[mt][r], 0.f), d = dA[mt][r];
          S[mt][r] = w0 * ex[mt][r] * (d - D) + 2.0f * w1 * rl * d;   // dS^T
        }
      // dQ^T[e][n] = sum_m K[m][e] dS^T[m][n]
      f32x16 O;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float av = Ks[(mt * 32 + rowmap(r)) * WB_LD + l31];
          O = __builtin_amdgcn_mfma_f32_32x32x2f32(av, S[mt][r], O, 0, 0, 0);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the LDS reads from being hoisted in bulk)
        }
      {
        float* __restrict__ ob = a.dq + (long)img * a.dqkv_bs + (long)(h * 32) * P;
        const unsigned oo = (unsigned)((long)(4 * half) * P + pixel(w, nt * 32 + l31)) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(ob + (long)((r & 3) + 8 * (r >> 2)) * P) + oo) = O[r] * a.scale;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // the statistics are in LDS
    // ---- "N" form: queries in the registers, the lane's column is a key; one half of the queries at a time ------
    f32x16 OV[2], OK[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { OV[mt][r] = 0.f; OK[mt][r] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f32x16 S[2], dA[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[mt][r] = 0.f; dA[mt][r] = 0.f; }
      const float* __restrict__ qs_frag = &Qs[(nt * 32 + l31) * WB_LD + half];
      const float* __restrict__ gs_frag = &Gs[(nt * 32 + l31) * WB_LD + half];
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          S[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qs_frag[2 * s], kf[mt][s], S[mt], 0, 0, 0);
          dA[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gs_frag[2 * s], vf[mt][s], dA[mt], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + rowmap(r);
        const int ny = n >> 3, nx = n & 7;
        const float mx = st[0][n], inv = st[1][n], D = st[2][n];
        const int rid_n = a.shift ? __shfl(rid_tok, n) : 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = mt * 32 + l31;
          float sc = S[mt][r] + ts[(ny - (m >> 3) + 7) * 15 + (nx - (m & 7) + 7)];
          if (a.shift) {
            const int rid_m = __shfl(rid_tok, m);
            if (rid_m != rid_n) sc += -100.0f;
          }
          const float p0 = fast_exp(sc - mx) * inv, rl = fmaxf(sc, 0.f), d = dA[mt][r];
          S[mt][r] = w0 * p0 + w1 * rl * rl;                        // A[n][m]
          dA[mt][r] = w0 * p0 * (d - D) + 2.0f * w1 * rl * d;        // dS[n][m]
        }
      }
      // dV^T[e][m] += sum_n dO[n][e] A[n][m];  dK^T[e][m] += sum_n (scale q)[n][e] dS[n][m];  dS also to LDS
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nt * 32 + rowmap(r);
        const float ga = Gs[n * WB_LD + l31], qa = Qs[n * WB_LD + l31];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          OV[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, S[mt][r], OV[mt], 0, 0, 0);
          OK[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa, dA[mt][r], OK[mt], 0, 0, 0);
          T[n * WB_TLD + mt * 32 + l31] = dA[mt][r];
        }
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float* __restrict__ dvb = a.dv + (long)img * a.dqkv_bs + (long)(h * 32) * P;
      float* __restrict__ dkb = a.dk + (long)img_k * a.dqkv_bs + (long)(hk * 32) * P;
      const unsigned ov = (unsigned)((long)(4 * half) * P + pixel(w, mt * 32 + l31)) * 4u;
      const unsigned ok = (unsigned)((long)(4 * half) * P + pixel(wk, mt * 32 + l31)) * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        *reinterpret_cast<float*>(reinterpret_cast<char*>(dvb + (long)((r & 3) + 8 * (r >> 2)) * P) + ov) = OV[mt][r];
        *reinterpret_cast<float*>(reinterpret_cast<char*>(dkb + (long)((r & 3) + 8 * (r >> 2)) * P) + ok) = OK[mt][r];
      }
    }
    // ---- table gradient: the dS tile is in LDS, one lane per table entry ----------------------------------------------
    __syncthreads();
    // table entry i = (dy + 7) * 15 + (dx + 7) collects dS[n][m] of every pair with n - m = (dy, dx)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int i = lane + 64 * t4;
      if (i < 225) {
        const int dy = i / 15 - 7, dx = i % 15 - 7;
        float sacc = 0.f;
        for (int ny = max(0, dy); ny <= min(7, 7 + dy); ++ny)
          for (int nx = max(0, dx); nx <= min(7, 7 + dx); ++nx)
            sacc += T[(ny * 8 + nx) * WB_TLD + (ny - dy) * 8 + (nx - dx)];
        tacc[t4] += sacc;
      }
    }
  }
  __syncthreads();
  const long slot = (long)h * a.chunks + chunk;
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
    if (lane + 64 * t4 < 225) a.part_table[slot * 225 + lane + 64 * t4] = (double)tacc[t4];
  // the two sums over the wave, in lane order
  __shared__ double wred[2][64];
  wred[0][lane] = g0;
  wred[1][lane] = g1;
  __syncthreads();
  if (lane < 2) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += wred[lane][i];
    a.part_w[slot * 2 + lane] = t;
  }
}

// dtable[i][h] (+)= sum_chunk part_table[h][chunk][i]: one workgroup per table entry; the last two workgroups
// add up the mixing-weight sums G0, G1 over every (head, chunk) slot into gsum[2].  Fixed tree: reproducible.
__global__ __launch_bounds__(256) void win_attn_bwd_reduce_kernel(const double* __restrict__ part_table,
                                                                  const double* __restrict__ part_w, int heads,
                                                                  int chunks, float* __restrict__ dtable,
                                                                  double* __restrict__ gsum, int accumulate) {
  __shared__ double red[256];
  const int e = blockIdx.x;
  double t = 0.0;
  if (e < 225 * heads) {
    const int i = e / heads, h = e - i * heads;
    for (int c = threadIdx.x; c < chunks; c += 256) t += part_table[((long)h * chunks + c) * 225 + i];
    t = block_sum_d(t, red);
    if (threadIdx.x == 0) dtable[e] = accumulate ? dtable[e] + (float)t : (float)t;
  } else {
    const int which = e - 225 * heads;
    for (long s = threadIdx.x; s < (long)heads * chunks; s += 256) t += part_w[s * 2 + which];
    t = block_sum_d(t, red);
    if (threadIdx.x == 0) gsum[which] = t;
  }
}
// d(w) through the two-way softmax of the mixing weights: s = softmax(w), dw_i = s_i (G_i - sum_j G_j s_j)
__global__ void win_attn_bwd_w_kernel(const double* __restrict__ gsum, const float* __restrict__ w2,
                                      float* __restrict__ dw2, int accumulate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double G0 = gsum[0], G1 = gsum[1];
    const double e0 = exp((double)w2[0]), e1 = exp((double)w2[1]);
    const double s0 = e0 / (e0 + e1), s1 = e1 / (e0 + e1);
    const double dot = G0 * s0 + G1 * s1;
    const float d0 = (float)(s0 * (G0 - dot)), d1 = (float)(s1 * (G1 - dot));
    dw2[0] = accumulate ? dw2[0] + d0 : d0;
    dw2[1] = accumulate ? dw2[1] + d1 : d1;
  }
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

// Scratch of the training operators (packed weights, partial sums): one set of buffers per device, grown on
// demand and kept until the process ends, so a step allocates nothing once the shapes have been seen.  All
// training calls on one device share it and order their use through the stream: calls that overlap on one
// device (two streams, two threads) race.
enum { SC_PACK = 0, SC_PART_A = 1, SC_PART_B = 2, SC_SMALL = 3, SC_SPLITK = 4, SC_SLOTS = 5 };
// Counts the reallocations of any scratch buffer of any device: a captured step holds the addresses it was recorded
// with, so a holder compares this number before every replay (jirf_train_scratch_generation).
std::atomic<long> g_scratch_generation{0};
void* train_scratch(int slot, size_t bytes, hipStream_t stream) {
  static std::mutex mu;
  static DeviceBuffer pool[64][SC_SLOTS];
  int dev = 0;
  JIRF_HIP_CHECK(hipGetDevice(&dev));
  JIRF_REQUIRE(dev >= 0 && dev < 64, JIRF_EINVAL, "training scratch: device %d", dev);
  std::lock_guard<std::mutex> lock(mu);
  DeviceBuffer& b = pool[dev][slot];
  if (b.bytes < bytes) {
    // Growing means a device synchronisation and a free / allocate pair, none of which a stream capture allows:
    // refuse before touching anything, so that the capture ends with a clear error and the buffers stay valid.
    hipStreamCaptureMode relaxed = hipStreamCaptureModeRelaxed;
    JIRF_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&relaxed));      // the query itself must not break a capture
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const hipError_t qe = hipStreamIsCapturing(stream, &st);
    JIRF_HIP_CHECK(hipThreadExchangeStreamCaptureMode(&relaxed));
    JIRF_REQUIRE(qe != hipSuccess || st == hipStreamCaptureStatusNone, JIRF_ESTATE,
                 "training scratch slot %d would have to grow to %zu bytes inside a stream capture: run one step at "
                 "this shape outside the capture first", slot, bytes);
    JIRF_HIP_CHECK(hipDeviceSynchronize());   // the old buffer may still be in use by queued kernels
    b.ensure(bytes + bytes / 4);
    g_scratch_generation.fetch_add(1);
  }
  return b.p;
}

inline long dense(long bs, long c, long p) { return bs ? bs : c * p; }

// Geometry of the packed form of one (weight, kind) pair: the GEMM's M, K, taps and the padded matrix.
struct PackGeom {
  int kind, M, K, taps, Kp, Mp;
  long floats() const { return (long)Kp * Mp + Mp; }   // the matrix and room for an expanded bias
};
PackGeom pack_geom(int w_kind, int Cin, int Cout, int ks) {
  PackGeom g{};
  g.kind = w_kind == JIRF_W_FWD ? 0 : w_kind == JIRF_W_BWD_DATA ? 1 : w_kind == JIRF_W_DECONV ? 2 : 4;
  g.taps = w_kind == JIRF_W_DECONV ? 1 : w_kind == JIRF_W_DOWN_BWD_DATA ? 9 : ks * ks;
  g.M = (w_kind == JIRF_W_DECONV || w_kind == JIRF_W_DOWN_BWD_DATA) ? 4 * Cout : Cout;
  g.K = g.taps * Cin;
  g.Mp = (g.M + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
  g.Kp = (g.K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  return g;
}
struct PackPlan {
  int device = 0, n = 0;
  unsigned blocks = 0;
  DeviceBuffer items;
  double bytes = 0.0;
};

// Per-kernel device time of the training operators (HIP events around every launch), for the calling thread.
thread_local KernelClock g_train_clock;

template <bool CONV, int WM, int WK, int TI, int TJ>
void dw_shared_launch(const float* dy, long dy_bs, const float* x, long x_bs, float*& part, double*& dbpart, bool want_db,
                      int& slabs, int B, int M, int K, int P, const ConvGeom& cg, hipStream_t s) {
  constexpr int TM = WM * TI * 32, TK = WK * TJ * 32;
  const int tm = (M + TM - 1) / TM, tk = (K + TK - 1) / TK;
  const long chunks = (long)((P + NT_PX - 1) / NT_PX) * B;
  // three workgroups per CU and then some, at least 8 chunks each (the slabs cost a round trip of M K floats each)
  slabs = (int)std::max<long>(1, std::min<long>(chunks / 8, (1024 + tm * tk - 1) / (tm * tk)));
  const int cps = (int)((chunks + slabs - 1) / slabs);
  slabs = (int)((chunks + cps - 1) / cps);
  part = static_cast<float*>(train_scratch(SC_PART_A, (size_t)slabs * M * K * sizeof(float), s));
  dbpart = want_db ? static_cast<double*>(train_scratch(SC_PART_B, (size_t)slabs * M * sizeof(double), s)) : nullptr;
  hipLaunchKernelGGL((dw_shared_kernel<CONV, WM, WK, TI, TJ>), dim3(tm, tk, slabs), dim3(256), 0, s, dy, dy_bs, x, x_bs,
                     part, dbpart, B, M, K, P, cps, cg);
}

void conv_dw_launch(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, float* db, int B, int M, int K,
                    int P, const ConvGeom* cg, int accumulate, hipStream_t s) {
  Scope sc(intern_label("train_dw M%d K%d P%d", M, K, P), 2.0 * M * (double)K * P * B,
           4.0 * B * (double)P * (M + (cg ? K / (cg->ks * cg->ks) : K)), s);
  float* part = nullptr;
  double* dbpart = nullptr;
  int slabs = 1;
  const ConvGeom geom = cg ? *cg : ConvGeom{};
#define JIRF_DW_SHARED(WM, WK, TI, TJ)                                                                             \
  (cg ? dw_shared_launch<true, WM, WK, TI, TJ>(dy, dy_bs, x, x_bs, part, dbpart, db != nullptr, slabs, B, M, K, P, \
                                               geom, s)                                                           \
      : dw_shared_launch<false, WM, WK, TI, TJ>(dy, dy_bs, x, x_bs, part, dbpart, db != nullptr, slabs, B, M, K, P, \
                                                geom, s))
  if (M > 64 && K > 64) {
    JIRF_DW_SHARED(2, 2, 2, 2);          // 128 x 128
  } else if (M > 64 && K > 32) {
    JIRF_DW_SHARED(4, 1, 2, 2);          // 256 x 64
  } else if (M > 64) {
    JIRF_DW_SHARED(4, 1, 2, 1);          // 256 x 32
  } else if (K > 64 && M > 32) {
    JIRF_DW_SHARED(1, 4, 2, 2);          // 64 x 256
  } else if (K > 64) {
    JIRF_DW_SHARED(1, 4, 1, 2);          // 32 x 256
  } else {
    // at most 64 x 64: every wave takes its own pixel chunks of the one tile row / column
    const int tm = (M + NT_T - 1) / NT_T, tk = (K + NT_T - 1) / NT_T;
    const long chunks = (long)((P + NT_PX - 1) / NT_PX) * B;
    // enough slabs to fill the chip (two workgroups per CU and then some), at least 16 chunks (4 per wave) each
    slabs = (int)std::max<long>(1, std::min<long>(chunks / 16, (2048 + tm * tk - 1) / (tm * tk)));
    const int cps = (int)((chunks + slabs - 1) / slabs);
    slabs = (int)((chunks + cps - 1) / cps);
    part = static_cast<float*>(train_scratch(SC_PART_A, (size_t)slabs * M * K * sizeof(float), s));
    dbpart = db ? static_cast<double*>(train_scratch(SC_PART_B, (size_t)slabs * M * sizeof(double), s)) : nullptr;
    if (cg)
      hipLaunchKernelGGL(linear_dw_kernel<true>, dim3(tm, tk, slabs), dim3(256), 0, s, dy, dy_bs, x, x_bs, part, dbpart,
                         B, M, K, P, cps, geom);
    else
      hipLaunchKernelGGL(linear_dw_kernel<false>, dim3(tm, tk, slabs), dim3(256), 0, s, dy, dy_bs, x, x_bs, part, dbpart,
                         B, M, K, P, cps, geom);
  }
#undef JIRF_DW_SHARED
  const long n = (long)M * K;
  // one launch adds the slabs of dW and of db
  const int vec = (n % 4 == 0 && n >= 32768 && slabs <= 64 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) ? 1 : 0;
  const unsigned w_blocks = (unsigned)(vec ? (n / 4 + 255) / 256 : (n + 15) / 16);
  hipLaunchKernelGGL(reduce_dw_kernel, dim3(w_blocks + (db ? (unsigned)((M + 15) / 16) : 0u)), dim3(256), 0, s, part, n,
                     slabs, dw, dbpart, M, db, w_blocks, vec, accumulate);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace
}  // namespace jirf

using namespace jirf;

extern "C" {

long jirf_train_scratch_generation(void) { return g_scratch_generation.load(); }

int jirf_train_kernel_timing(int enable) {
  return guarded_t([&] {
    g_train_clock.enabled = enable != 0;
    g_train_clock.begin_forward();
    set_current_clock(enable ? &g_train_clock : nullptr);
    return JIRF_OK;
  });
}

int jirf_train_kernel_times(const char** names_out, float* ms_out, double* flops_out, double* bytes_out, int* launches_out,
                            int capacity, int* n_out) {
  return guarded_t([&] {
    JIRF_REQUIRE(n_out, JIRF_EINVAL, "null argument");
    g_train_clock.collect();
    g_train_clock.begin_forward();
    const auto& agg = g_train_clock.agg;
    *n_out = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < capacity; ++i) {
      if (names_out) names_out[i] = agg[i].name;
      if (ms_out) ms_out[i] = agg[i].ms;
      if (flops_out) flops_out[i] = agg[i].flops;
      if (bytes_out) bytes_out[i] = agg[i].bytes;
      if (launches_out) launches_out[i] = agg[i].launches;
    }
    return JIRF_OK;
  });
}

int jirf_train_smooth_l1(const float* pred, const float* target, float* grad, long n, float loss_weight,
                         double* loss_dev, double* loss_host, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(pred && target && n > 0 && (loss_dev || loss_host), JIRF_EINVAL, "smooth_l1: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (int)std::min<long>(1024, (n + 255) / 256);
    double* pd = static_cast<double*>(train_scratch(SC_SMALL, (size_t)(blocks + 1) * sizeof(double), s));
    {
      Scope sc("train_smooth_l1", 6.0 * n, 12.0 * n, s);
      hipLaunchKernelGGL(smooth_l1_kernel, dim3(blocks), dim3(256), 0, s, pred, target, grad, n,
                         loss_weight / (float)n, pd);
      hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, s, pd, blocks, pd + blocks,
                         (double)loss_weight / (double)n, loss_dev);
    }
    JIRF_HIP_CHECK(hipGetLastError());
    if (loss_host) {
      JIRF_HIP_CHECK(hipMemcpyAsync(loss_host, pd + blocks, sizeof(double), hipMemcpyDeviceToHost, s));
      JIRF_HIP_CHECK(hipStreamSynchronize(s));
    }
    return JIRF_OK;
  });
}

int jirf_train_sumsq(const float* x, long n, double* out_dev, double* out_host, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && n > 0 && (out_dev || out_host), JIRF_EINVAL, "sumsq: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = (int)std::min<long>(1024, (n + 255) / 256);
    double* pd = static_cast<double*>(train_scratch(SC_SMALL, (size_t)(blocks + 1) * sizeof(double), s));
    {
      Scope sc("train_grad_sumsq", 2.0 * n, 4.0 * n, s);
      hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, x, n, pd);
      hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, s, pd, blocks, out_dev ? out_dev : pd + blocks, 1.0,
                         static_cast<double*>(nullptr));
    }
    JIRF_HIP_CHECK(hipGetLastError());
    if (out_host) {
      JIRF_HIP_CHECK(hipMemcpyAsync(out_host, out_dev ? out_dev : pd + blocks, sizeof(double), hipMemcpyDeviceToHost, s));
      JIRF_HIP_CHECK(hipStreamSynchronize(s));
    }
    return JIRF_OK;
  });
}

int jirf_train_adam_step(float* p, const float* g, float* m, float* v, long n, const double* sumsq_dev, float max_norm,
                         float grad_scale, float lr, float beta0, float beta1, float eps, int step, int torch_form,
                         void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(p && g && m && v && n > 0 && step >= 1 && grad_scale > 0.f, JIRF_EINVAL, "adam_step: bad arguments");
    JIRF_REQUIRE(max_norm <= 0.f || sumsq_dev, JIRF_EINVAL, "adam_step: clipping needs the gradient's sum of squares");
    const float bc0 = 1.0f - std::pow(beta0, (float)step), bc1 = 1.0f - std::pow(beta1, (float)step);
    const int blocks = (int)std::min<long>(4096, (n + 255) / 256);
    Scope sc("train_adam", 12.0 * n, 28.0 * n, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                       sumsq_dev, max_norm, grad_scale, lr, beta0, beta1, eps, bc0, bc1, torch_form);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_conv(const float* x, long x_bs, const float* w, int w_kind, const float* bias, const float* r, long r_bs,
                    float* y, long y_bs, int B, int Cin, int Cout, int Hin, int Win, int ks, int stride, int pad,
                    int epi, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, JIRF_EINVAL, "train_conv: bad arguments");
    const bool packed = (w_kind & JIRF_W_PACKED) != 0;
    w_kind &= ~JIRF_W_PACKED;
    JIRF_REQUIRE(w_kind == JIRF_W_FWD || w_kind == JIRF_W_BWD_DATA || w_kind == JIRF_W_DECONV ||
                     w_kind == JIRF_W_DOWN_BWD_DATA, JIRF_EINVAL, "train_conv: weight kind %d", w_kind);
    hipStream_t s = static_cast<hipStream_t>(stream);
    GemmArgs g{};
    int loader = LD_PLAIN, M = Cout, taps = 1, pk_cin = Cin;
    g.B = B; g.imgs_per_group = 1; g.epi = epi;
    if (w_kind == JIRF_W_DECONV) {
      JIRF_REQUIRE(epi == 0 && !r, JIRF_EINVAL, "train_conv: the transposed convolution takes no epilogue");
      M = 4 * Cout; g.K = Cin; g.P = Hin * Win; g.H = Hin; g.W = Win; g.epi = EP_SHUFFLE;
    } else if (w_kind == JIRF_W_DOWN_BWD_DATA) {
      JIRF_REQUIRE(epi == 0 && !r, JIRF_EINVAL, "train_conv: the strided data gradient takes no epilogue");
      M = 4 * Cout; taps = 9; g.K = 9 * Cin; g.P = Hin * Win; g.H = Hin; g.W = Win; g.Hin = Hin; g.Win = Win;
      g.Cin = Cin; g.ks = 3; g.stride = 1; g.pad = 1; g.epi = EP_SHUFFLE; loader = LD_CONV3;
    } else {
      JIRF_REQUIRE(ks >= 1 && ks <= 7 && stride >= 1 && pad >= 0, JIRF_EINVAL, "train_conv: geometry k%d s%d p%d", ks, stride, pad);
      taps = ks * ks;
      int p_eff = pad;
      if (w_kind == JIRF_W_BWD_DATA) {
        JIRF_REQUIRE(stride == 1 && pad <= ks - 1, JIRF_EINVAL, "train_conv: the data gradient form needs stride 1");
        p_eff = ks - 1 - pad;
      }
      g.K = taps * Cin;
      if (ks == 1 && stride == 1 && pad == 0) {
        g.P = Hin * Win;
      } else {
        loader = LD_CONV3;
        g.H = (Hin + 2 * p_eff - ks) / stride + 1;
        g.W = (Win + 2 * p_eff - ks) / stride + 1;
        g.P = g.H * g.W; g.Hin = Hin; g.Win = Win; g.Cin = Cin; g.ks = ks; g.stride = stride; g.pad = p_eff;
      }
    }
    (void)pk_cin;
    const PackGeom pg = pack_geom(w_kind, Cin, Cout, ks);
    JIRF_REQUIRE(pg.M == M && pg.K == g.K && pg.taps == taps, JIRF_EINVAL, "train_conv: packed geometry");
    const int Mp = pg.Mp, Kp = pg.Kp;
    const float* wt = w;
    const float* bias_use = bias;
    if (packed) {
      // w is the packed form already (jirf_train_pack_plan_run); the expanded bias of a transposed convolution
      // sits behind the matrix
      if (bias && w_kind == JIRF_W_DECONV) bias_use = w + (size_t)Kp * Mp;
    } else {
      float* wp = static_cast<float*>(train_scratch(SC_PACK, (size_t)pg.floats() * sizeof(float), s));
      {
        Scope sc(intern_label("train_pack_weights K%d M%d", g.K, M), 0.0, 4.0 * ((double)Kp * Mp + (double)g.K * M), s);
        hipLaunchKernelGGL(pack_train_kernel, dim3((unsigned)(((long)Kp * Mp + 255) / 256)), dim3(256), 0, s, w, wp,
                           pg.kind, M, Cin, taps, g.K, Kp, Mp);
      }
      if (bias && w_kind == JIRF_W_DECONV) {
        float* b4 = wp + (size_t)Kp * Mp;
        hipLaunchKernelGGL(expand_bias4_kernel, dim3((M + 255) / 256), dim3(256), 0, s, bias, b4, M);
        bias_use = b4;
      }
      JIRF_HIP_CHECK(hipGetLastError());
      wt = wp;
    }
    g.M = M; g.Mpad = Mp; g.wt = wt; g.bias = bias_use;
    // slabs of the split-K form (deep levels: long reductions over a handful of pixels)
    g.splitk_ws = static_cast<float*>(train_scratch(SC_SPLITK, SPLITK_WS_FLOATS * sizeof(float), s));
    g.splitk_floats = SPLITK_WS_FLOATS;
    const long Pin = (long)Hin * Win;
    const long Pout = (g.epi == EP_SHUFFLE) ? 4L * g.P : g.P;
    g.x = tv(const_cast<float*>(x), dense(x_bs, Cin, Pin));
    g.y = tv(y, dense(y_bs, Cout, Pout));
    if (r) g.r = tv(const_cast<float*>(r), dense(r_bs, Cout, Pout));
    JIRF_REQUIRE((g.epi == EP_RESID || g.epi == EP_GELU_MUL) == (r != nullptr), JIRF_EINVAL,
                 "train_conv: epilogue %d and the residual / gate operand do not match", epi);
    launch_gemm(g, loader, s);
    return JIRF_OK;
  });
}

long jirf_train_packed_floats(int w_kind, int Cin, int Cout, int ks) {
  if (Cin <= 0 || Cout <= 0 || ks < 1 || (w_kind != JIRF_W_FWD && w_kind != JIRF_W_BWD_DATA && w_kind != JIRF_W_DECONV &&
                                           w_kind != JIRF_W_DOWN_BWD_DATA))
    return -1;
  return pack_geom(w_kind, Cin, Cout, ks).floats();
}

int jirf_train_pack_plan_create(int n, const float* const* w, const float* const* bias, float* const* wt,
                                const int* w_kind, const int* Cin, const int* Cout, const int* ks, void** plan_out) {
  return guarded_t([&] {
    JIRF_REQUIRE(n > 0 && w && wt && w_kind && Cin && Cout && ks && plan_out, JIRF_EINVAL, "pack_plan_create: bad arguments");
    std::vector<PackItem> items((size_t)n);
    unsigned blocks = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
      JIRF_REQUIRE(w[i] && wt[i] && jirf_train_packed_floats(w_kind[i], Cin[i], Cout[i], ks[i]) > 0, JIRF_EINVAL,
                   "pack_plan_create: item %d", i);
      const PackGeom g = pack_geom(w_kind[i], Cin[i], Cout[i], ks[i]);
      PackItem& it = items[(size_t)i];
      it.w = w[i];
      it.bias = (bias && w_kind[i] == JIRF_W_DECONV) ? bias[i] : nullptr;
      it.wt = wt[i];
      it.kind = g.kind; it.M = g.M; it.Cin = Cin[i]; it.taps = g.taps; it.K = g.K; it.Kp = g.Kp; it.Mp = g.Mp;
      it.block0 = blocks;
      it.tiles_o = (unsigned)((g.Mp + 63) / 64);
      blocks += it.tiles_o * (unsigned)((g.Kp + 63) / 64);
      bytes += 4.0 * ((double)g.Kp * g.Mp + (double)g.K * g.M);
    }
    auto plan = std::make_unique<PackPlan>();
    JIRF_HIP_CHECK(hipGetDevice(&plan->device));
    plan->n = n;
    plan->blocks = blocks;
    plan->bytes = bytes;
    plan->items.ensure(items.size() * sizeof(PackItem));
    JIRF_HIP_CHECK(hipMemcpy(plan->items.p, items.data(), items.size() * sizeof(PackItem), hipMemcpyHostToDevice));
    *plan_out = plan.release();
    return JIRF_OK;
  });
}

int jirf_train_pack_plan_run(void* plan, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(plan, JIRF_EINVAL, "pack_plan_run: null plan");
    const PackPlan* pl = static_cast<const PackPlan*>(plan);
    int dev = 0;
    JIRF_HIP_CHECK(hipGetDevice(&dev));
    JIRF_REQUIRE(dev == pl->device, JIRF_EINVAL, "pack_plan_run: the plan belongs to device %d, current device %d", pl->device, dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(intern_label("train_pack_plan n%d", pl->n), 0.0, pl->bytes, s);
    hipLaunchKernelGGL(pack_plan_kernel, dim3(pl->blocks), dim3(256), 0, s, static_cast<const PackItem*>(pl->items.p), pl->n);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

void jirf_train_pack_plan_destroy(void* plan) { delete static_cast<PackPlan*>(plan); }

int jirf_train_conv_bwd_weight(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, float* db, int B,
                               int M, int Cin, int Hout, int Wout, int Hin, int Win, int ks, int stride, int pad,
                               int accumulate, void* stream) {
  return guarded_t([&] {
    const int P = Hout * Wout;
    JIRF_REQUIRE(dy && x && dw && B > 0 && M > 0 && Cin > 0 && P >= 16 && P % 16 == 0 && Wout >= 4 && ks >= 1 &&
                     stride >= 1 && pad >= 0,
                 JIRF_EINVAL,
                 "conv_bwd_weight: bad arguments (the output plane must be a multiple of 16 pixels, at least 4 wide)");
    JIRF_REQUIRE((Hin + 2 * pad - ks) / stride + 1 == Hout && (Win + 2 * pad - ks) / stride + 1 == Wout, JIRF_EINVAL,
                 "conv_bwd_weight: geometry out %dx%d in %dx%d k%d s%d p%d", Hout, Wout, Hin, Win, ks, stride, pad);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool plain = ks == 1 && stride == 1 && pad == 0;
    ConvGeom cg{ks, stride, pad, Hin, Win, Wout};
    JIRF_REQUIRE((dense(dy_bs, M, P) % 4) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                     (!plain || ((dense(x_bs, Cin, P) % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)),
                 JIRF_EINVAL, "conv_bwd_weight: operands must be 16-byte aligned");
    conv_dw_launch(dy, dense(dy_bs, M, P), x, dense(x_bs, Cin, (long)Hin * Win), dw, db, B, M, Cin * ks * ks, P,
                   plain ? nullptr : &cg, accumulate, s);
    return JIRF_OK;
  });
}

int jirf_train_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, int B, int M, int K, int P,
                                 int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(dy && x && dw && B > 0 && M > 0 && K > 0 && P >= 4 && P % 4 == 0, JIRF_EINVAL,
                 "linear_bwd_weight: bad arguments (P must be a multiple of 4)");
    conv_dw_launch(dy, (long)M * P, x, (long)K * P, dw, db, B, M, K, P, nullptr, accumulate,
                   static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_train_layernorm(const float* x, long x_bs, const float* w, const float* b, float* y, long y_bs, int B, int C,
                         int P, int kind, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && w && y && B > 0 && C > 0 && P > 0 && (kind == 1 || kind == 2), JIRF_EINVAL, "train_layernorm: bad arguments");
    launch_layernorm(tv(const_cast<float*>(x), dense(x_bs, C, P)), w, b, tv(y, dense(y_bs, C, P)), B, C, P, kind,
                     static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_train_layernorm_bwd(const float* x, long x_bs, const float* dy, long dy_bs, const float* w, const float* add,
                             long add_bs, float* dx, long dx_bs, float* dw, float* db, int B, int C, int P, int kind,
                             int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && dy && w && dx && dw && B > 0 && C > 0 && P > 0 && (kind == 1 || kind == 2), JIRF_EINVAL,
                 "layernorm_bwd: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int gx = (P + 63) / 64;
    const int slots = gx * B;
    double* pw = static_cast<double*>(train_scratch(SC_PART_A, (size_t)slots * C * sizeof(double), s));
    double* pb = static_cast<double*>(train_scratch(SC_PART_B, (size_t)slots * C * sizeof(double), s));
    Scope sc(intern_label("train_layernorm_bwd C%d P%d", C, P), 12.0 * B * C * (double)P,
             4.0 * B * C * (double)P * (add ? 4 : 3), s);
    if (slots < 512)
      hipLaunchKernelGGL(layernorm_bwd_kernel<16>, dim3(gx, B), dim3(1024), 0, s, x, dense(x_bs, C, P), dy,
                         dense(dy_bs, C, P), w, add, dense(add_bs, C, P), dx, dense(dx_bs, C, P), pw, pb, C, P,
                         kind == 1 ? 1 : 0);
    else
      hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(gx, B), dim3(256), 0, s, x, dense(x_bs, C, P), dy,
                         dense(dy_bs, C, P), w, add, dense(add_bs, C, P), dx, dense(dx_bs, C, P), pw, pb, C, P,
                         kind == 1 ? 1 : 0);
    hipLaunchKernelGGL(reduce_cols_d_kernel, dim3(C, (db && kind == 1) ? 2 : 1), dim3(256), 0, s, pw, dw, pb, db, C, slots,
                       accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_gelu(const float* pre, long pre_bs, const float* x2, long x2_bs, float* y, long y_bs, int B, long n_img,
                    int mode, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(pre && y && B > 0 && n_img > 0 && (mode == 0 || ((mode == 1 || mode == 3) && x2)), JIRF_EINVAL, "train_gelu: bad arguments");
    const int gx = (int)std::min<long>(2048, (n_img + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(intern_label("train_gelu n%ld", n_img), 10.0 * B * (double)n_img, 4.0 * B * (double)n_img * (2 + (mode != 0)), s);
    const long pbs = pre_bs ? pre_bs : n_img, xbs = x2_bs ? x2_bs : n_img, ybs = y_bs ? y_bs : n_img;
    auto al = [](const void* p, long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && bs % 4 == 0; };
    if (n_img % 4 == 0 && al(pre, pbs) && al(y, ybs) && (!x2 || al(x2, xbs))) {
      const int gx4 = (int)std::min<long>(2048, (n_img / 4 + 255) / 256);
      hipLaunchKernelGGL(gelu_fwd4_kernel, dim3(gx4, B), dim3(256), 0, s, pre, pbs, x2, xbs, y, ybs, n_img / 4, mode);
    } else {
      hipLaunchKernelGGL(gelu_fwd_kernel, dim3(gx, B), dim3(256), 0, s, pre, pbs, x2, xbs, y, ybs, n_img, mode);
    }
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_gelu_bwd(const float* pre, long pre_bs, const float* x2, long x2_bs, const float* dy, long dy_bs,
                        float* dxa, long dxa_bs, float* dxb, long dxb_bs, int B, long n_img, int mode, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(pre && dy && dxa && B > 0 && n_img > 0 && (mode == 0 || mode == 2 || ((mode == 1 || mode == 3) && x2 && dxb)), JIRF_EINVAL,
                 "gelu_bwd: bad arguments");
    const int gx = (int)std::min<long>(2048, (n_img + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(intern_label("train_gelu_bwd n%ld", n_img), 20.0 * B * (double)n_img, 4.0 * B * (double)n_img * (3 + 2 * (mode == 1 || mode == 3)), s);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(gx, B), dim3(256), 0, s, pre, pre_bs ? pre_bs : n_img, x2,
                       x2_bs ? x2_bs : n_img, dy, dy_bs ? dy_bs : n_img, dxa, dxa_bs ? dxa_bs : n_img, dxb,
                       dxb_bs ? dxb_bs : n_img, n_img, mode);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_add(const float* a, long a_bs, const float* b, long b_bs, float* y, long y_bs, int B, long n_img,
                   void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(a && y && B > 0 && n_img > 0, JIRF_EINVAL, "train_add: bad arguments");
    const int gx = (int)std::min<long>(2048, (n_img + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(intern_label("train_add n%ld", n_img), 1.0 * B * (double)n_img, 4.0 * B * (double)n_img * (b ? 3 : 2), s);
    hipLaunchKernelGGL(add_kernel, dim3(gx, B), dim3(256), 0, s, a, a_bs ? a_bs : n_img, b, b_bs ? b_bs : n_img,
                       static_cast<const float*>(nullptr), y, y_bs ? y_bs : n_img, n_img);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_scale_add(const float* a, long a_bs, const float* b, long b_bs, const float* scale, float* y, long y_bs,
                         int B, long n_img, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(b && scale && y && B > 0 && n_img > 0, JIRF_EINVAL, "train_scale_add: bad arguments");
    const int gx = (int)std::min<long>(2048, (n_img + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scope sc(intern_label("train_scale_add n%ld", n_img), 2.0 * B * (double)n_img, 4.0 * B * (double)n_img * (a ? 3 : 2), s);
    hipLaunchKernelGGL(add_kernel, dim3(gx, B), dim3(256), 0, s, a, a_bs ? a_bs : n_img, b, b_bs ? b_bs : n_img, scale, y,
                       y_bs ? y_bs : n_img, n_img);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_drop_path_scales(float* out, const float* rate, int rows, int B, unsigned long long seed,
                                unsigned long long* counter, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(out && rate && counter && rows > 0 && B > 0, JIRF_EINVAL, "drop_path_scales: bad arguments");
    hipLaunchKernelGGL(drop_scales_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), out, rate, rows, B, seed,
                       counter);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_channel_sum(const float* x, long x_bs, float* out, int B, int C, long P, int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && out && B > 0 && C > 0 && P > 0, JIRF_EINVAL, "channel_sum: bad arguments");
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       dense(x_bs, C, P), out, B, P, accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_dwconv3x3(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs, int B, int C,
                         int H, int W, int flip, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "train_dwconv3x3: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* wu = w;
    if (flip) {
      float* wf = static_cast<float*>(train_scratch(SC_SMALL, (size_t)C * 9 * sizeof(float), s));
      hipLaunchKernelGGL(flip9_kernel, dim3((C * 9 + 255) / 256), dim3(256), 0, s, w, wf, C);
      JIRF_HIP_CHECK(hipGetLastError());
      wu = wf;
    }
    const long P = (long)H * W;
    launch_dwconv3x3(tv(const_cast<float*>(x), dense(x_bs, C, P)), wu, bias, tv(y, dense(y_bs, C, P)), B, C, H, W, 0, s);
    return JIRF_OK;
  });
}

int jirf_train_dwconv3x3_bwd_weight(const float* x, long x_bs, const float* dy, long dy_bs, float* dw, float* db, int B,
                                    int C, int H, int W, int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(x && dy && dw && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "dwconv3x3_bwd_weight: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long P = (long)H * W;
    // row slabs of at least 4096 pixels per workgroup (a thread's fp32 sums stay short), enough workgroups to fill the chip
    const int slabs = (int)std::max<long>(1, std::min<long>(std::min<long>(P / 4096, H), std::max<long>((4096 + (long)B * C - 1) / ((long)B * C), P / 65536)));
    double* part = static_cast<double*>(train_scratch(SC_PART_A, (size_t)B * slabs * C * 10 * sizeof(double), s));
    Scope sc(intern_label("train_dwconv_dw C%d H%d W%d", C, H, W), 20.0 * B * C * (double)P, 8.0 * B * C * (double)P, s);
    hipLaunchKernelGGL(dwconv_dw_kernel, dim3(C, B * slabs), dim3(256), 0, s, x, dense(x_bs, C, P), dy, dense(dy_bs, C, P),
                       part, C, H, W, slabs);
    hipLaunchKernelGGL(dwconv_dw_reduce_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, s, part, B * slabs, C, dw, db,
                       accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

int jirf_train_window_attention(const float* qkv, long qkv_bs, const float* table, const float* w2, float* out,
                                long out_bs, int B, int C, int H, int W, int heads, int shift, int coupled, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(qkv && table && w2 && out && B > 0, JIRF_EINVAL, "train_window_attention: bad arguments");
    const long P = (long)H * W;
    WinAttnArgs wa{};
    wa.B = B; wa.C = C; wa.H = H; wa.W = W; wa.heads = heads; wa.shift = shift; wa.coupled = coupled ? 1 : 0;
    const TV t = tv(const_cast<float*>(qkv), dense(qkv_bs, 3 * C, P));
    wa.q = tv_chan(t, 0, P); wa.k = tv_chan(t, C, P); wa.v = tv_chan(t, 2 * C, P);
    wa.out = tv(out, dense(out_bs, C, P));
    wa.table = table; wa.scale = 1.0f / std::sqrt(32.0f); wa.w_dev = w2;
    launch_window_attention(wa, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_train_window_attention_bwd(const float* qkv, long qkv_bs, const float* table, const float* w2, const float* dout,
                                    long dout_bs, float* dqkv, long dqkv_bs, float* dtable, float* dw2, int B, int C, int H,
                                    int W, int heads, int shift, int coupled, int accumulate, void* stream) {
  return guarded_t([&] {
    JIRF_REQUIRE(qkv && table && w2 && dout && dqkv && dtable && dw2 && B > 0 && heads > 0 && C == heads * 32, JIRF_EINVAL,
                 "window_attention_bwd: bad arguments (head dim 32 only)");
    JIRF_REQUIRE(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8 && (shift == 0 || shift == 4), JIRF_EINVAL,
                 "window_attention_bwd: %dx%d shift %d", H, W, shift);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long P = (long)H * W;
    const int n_win = B * (H / 8) * (W / 8);
    WinBwdArgs a{};
    a.B = B; a.C = C; a.H = H; a.W = W; a.heads = heads; a.shift = shift; a.coupled = coupled ? 1 : 0;
    a.chunks = std::max(1, std::min(n_win, (2048 + heads - 1) / heads));
    a.scale = 1.0f / std::sqrt(32.0f);
    a.qkv_bs = dense(qkv_bs, 3 * C, P); a.dout_bs = dense(dout_bs, C, P); a.dqkv_bs = dense(dqkv_bs, 3 * C, P);
    a.q = qkv; a.k = qkv + (long)C * P; a.v = qkv + 2L * C * P; a.dout = dout;
    a.dq = dqkv; a.dk = dqkv + (long)C * P; a.dv = dqkv + 2L * C * P;
    a.table = table; a.w2 = w2;
    a.part_table = static_cast<double*>(train_scratch(SC_PART_A, (size_t)heads * a.chunks * 225 * sizeof(double), s));
    a.part_w = static_cast<double*>(train_scratch(SC_PART_B, ((size_t)heads * a.chunks * 2 + 2) * sizeof(double), s));
    double* gsum = a.part_w + (size_t)heads * a.chunks * 2;
    Scope sc(intern_label("train_win_attn_bwd C%d H%d W%d", C, H, W), 10.0 * 64 * C * (double)P * B, 28.0 * C * (double)P * B, s);
    hipLaunchKernelGGL(win_attn_bwd_kernel, dim3(a.chunks, heads), dim3(64), 0, s, a);
    hipLaunchKernelGGL(win_attn_bwd_reduce_kernel, dim3(225 * heads + 2), dim3(256), 0, s, a.part_table, a.part_w, heads,
                       a.chunks, dtable, gsum, accumulate);
    hipLaunchKernelGGL(win_attn_bwd_w_kernel, dim3(1), dim3(64), 0, s, gsum, w2, dw2, accumulate);
    JIRF_HIP_CHECK(hipGetLastError());
    return JIRF_OK;
  });
}

}  // extern "C"
