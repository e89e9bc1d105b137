// AST's sparse window attention (ASSA) for gfx950 — reference
// basicsr/models/archs/ast_arch.py:219-318 (WindowAttention_sparse), :768-837 (shift, partition,
// shift mask, reverse) and :33-49 (bmm_4d).
//
// One wave per (window, head): lane n is token n of the 8 x 8 window (64 tokens = 64 lanes), so a
// score row, its softmax, the ReLU^2 branch and the output row all live in one lane's registers
// and need no cross-lane traffic.  The K tile, then the V tile (64 tokens x 32 dims) is staged in the
// wave's LDS slot and read back as wave-uniform float4 broadcasts.  Roll, window partition and window reverse are pure
// addressing: a token reads and writes pixel ((wy*8 + ty + shift) mod H, (wx*8 + tx + shift) mod W).
//
// The reference flattens q as [window, head] but k as [head, window] before its batched matmul
// (bmm_4d), so score block j = b * heads + h multiplies q[b, h] with k[b', h'] where
// h' = j / B_, b' = j % B_  (B_ = number of windows in the call).  Reproduced here on purpose.
#include "common.h"

namespace jirf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WA_D = 32;       // head dim
constexpr int WA_N = 64;       // tokens per window
constexpr int WA_LD = 36;      // LDS row stride (floats): 16-byte aligned rows

__global__ __launch_bounds__(256, 3) void win_attn_kernel(const WinAttnArgs a, int n_tasks) {
  __shared__ __attribute__((aligned(16))) float KVs[4][WA_N * WA_LD];   // K during the scores, then V
  __shared__ float Ts[4][232];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // neighbouring windows share an XCD's L2
  if (task >= n_tasks) return;   // whole wave exits together; no block-wide barrier is used below
  const int H = a.H, W = a.W, heads = a.heads;
  const int nwx = W >> 3, nW = (H >> 3) * nwx;
  const long P = (long)H * W;
  // task = (global window) * heads + head, global window = image * nW + window
  const int gw = task / heads, h = task - gw * heads;
  const int img = gw / nW, w = gw - img * nW;
  // K comes from another (window, head): see the header comment
  const int span = a.coupled ? a.B * nW : nW;         // B_ of the reference call
  const int j = (a.coupled ? gw : w) * heads + h;
  const int hk = j / span, bk = j - hk * span;
  const int img_k = a.coupled ? bk / nW : img;
  const int wk = a.coupled ? bk - img_k * nW : bk;

  const int ty = lane >> 3, tx = lane & 7;
  auto pixel = [&](int win) {
    const int wy = win / nwx, wx = win - wy * nwx;
    int y = wy * 8 + ty + a.shift, x = wx * 8 + tx + a.shift;
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    return (long)y * W + x;
  };
  const long pq = pixel(w), pk = pixel(wk);

  float* __restrict__ ks = KVs[wave];
  float* __restrict__ vs = KVs[wave];
  float* __restrict__ ts = Ts[wave];
  // one LDS tile per wave (three workgroups fit a CU instead of two): K goes in now, V waits in registers
  // until the scores are done
  float vreg[WA_D];
  {
    const float* __restrict__ kp = a.k.p + (long)img_k * a.k.bs + (long)(hk * WA_D) * P + pk;
#pragma unroll
    for (int e = 0; e < WA_D; ++e) ks[lane * WA_LD + e] = kp[(long)e * P];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {   // 225 table entries: four unpredicated loads per lane, in flight together
      const int i = lane + 64 * t4;
      const float tv = a.table[min(i, 224) * heads + h];
      if (i < 225) ts[i] = tv;
    }
  }
  float q[WA_D];
  {
    const float* __restrict__ qp = a.q.p + (long)img * a.q.bs + (long)(h * WA_D) * P + pq;
#pragma unroll
    for (int e = 0; e < WA_D; ++e) q[e] = qp[(long)e * P] * a.scale;
    const float* __restrict__ vp = a.v.p + (long)img * a.v.bs + (long)(h * WA_D) * P + pq;
#pragma unroll
    for (int e = 0; e < WA_D; ++e) vreg[e] = vp[(long)e * P];
  }
  // region id of this token on the shifted grid (Swin shift mask), from the window position
  int rid = 0;
  if (a.shift) {
    const int wy = w / nwx, wx = w - wy * nwx;
    const int sy = wy * 8 + ty, sx = wx * 8 + tx;
    const int rh = sy < H - 8 ? 0 : (sy < H - a.shift ? 1 : 2);
    const int rw = sx < W - 8 ? 0 : (sx < W - a.shift ? 1 : 2);
    rid = 3 * rh + rw;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  float sc[WA_N];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < WA_N; ++m) {
    const float4* __restrict__ kr = reinterpret_cast<const float4*>(&ks[m * WA_LD]);
    float s = 0.f;
#pragma unroll
    for (int e4 = 0; e4 < WA_D / 4; ++e4) {
      const float4 kv = kr[e4];
      s = fmaf(q[4 * e4 + 0], kv.x, s);
      s = fmaf(q[4 * e4 + 1], kv.y, s);
      s = fmaf(q[4 * e4 + 2], kv.z, s);
      s = fmaf(q[4 * e4 + 3], kv.w, s);
    }
    // relative position bias: idx = (ny - my + 7) * 15 + (nx - mx + 7)
    const int my = m >> 3, mxx = m & 7;
    s += ts[(ty - my + 7) * 15 + (tx - mxx + 7)];
    if (a.shift) {
      const int rid_m = __shfl(rid, m);
      if (rid_m != rid) s += -100.0f;
    }
    sc[m] = s;
    mx = fmaxf(mx, s);
  }
  // every lane has read K: the tile now takes V
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < WA_D; ++e) vs[lane * WA_LD + e] = vreg[e];
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < WA_N; ++m) sum += fast_exp(sc[m] - mx);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float o[WA_D];
#pragma unroll
  for (int e = 0; e < WA_D; ++e) o[e] = 0.f;
#pragma unroll
  for (int m = 0; m < WA_N; ++m) {
    const float r = fmaxf(sc[m], 0.f);
    const float p = (fast_exp(sc[m] - mx) * inv) * a.w_softmax + (r * r) * a.w_relu2;
    const float4* __restrict__ vr = reinterpret_cast<const float4*>(&vs[m * WA_LD]);
#pragma unroll
    for (int e4 = 0; e4 < WA_D / 4; ++e4) {
      const float4 vv = vr[e4];
      o[4 * e4 + 0] = fmaf(p, vv.x, o[4 * e4 + 0]);
      o[4 * e4 + 1] = fmaf(p, vv.y, o[4 * e4 + 1]);
      o[4 * e4 + 2] = fmaf(p, vv.z, o[4 * e4 + 2]);
      o[4 * e4 + 3] = fmaf(p, vv.w, o[4 * e4 + 3]);
    }
  }
  float* __restrict__ op = a.out.p + (long)img * a.out.bs + (long)(h * WA_D) * P + pq;
#pragma unroll
  for (int e = 0; e < WA_D; ++e) op[(long)e * P] = o[e];
}

// ---- matrix-core form ------------------------------------------------------------------------------------
// The vector-ALU kernel above is bound by LDS bandwidth: every K / V row is a 16-byte broadcast read,
// 1024 of them per (window, head).  Here the two 64 x 64 x 32 products run on fp32 MFMA (32x32x2) with
// operands taken straight from registers:
//   S^T[m][n] = sum_e K[m][e] Q[n][e]      A = K fragments, B = Q fragments (both loaded from global in
//                                          fragment layout: lane l holds row / column l & 31, k = l >> 5)
// Computing the scores transposed puts one query n in a lane's column, so its softmax runs over that
// lane's 32 registers plus the partner lane (l ^ 32) and needs one cross-lane exchange.
//   O^T[e][n] = sum_m V[m][e] P[m][n]      B = the probability registers as they are: register r of lane
//                                          half h holds key m = 32 mt + (r & 3) + 8 (r >> 2) + 4 h, and the
//                                          reduction may visit the keys in any order as long as A follows:
//                                          A = V[m_h][e] read from the wave's LDS tile (lane = channel e).
constexpr int WA_LDV = 40;     // V tile row stride: rows m and m + 4 land on disjoint bank halves

__global__ __launch_bounds__(256, 3) void win_attn_mfma_kernel(const WinAttnArgs a, int n_tasks) {
  __shared__ float Vs[4][WA_N * WA_LDV];
  __shared__ float Ts[4][232];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l31 = lane & 31, half = lane >> 5;
  // The four waves of a workgroup take four windows that are neighbours along x, for the same head: a window row is
  // 32 bytes of a 128-byte line, and the four waves ask for the four quarters of each line at about the same time.
  const int H = a.H, W = a.W, heads = a.heads;
  const int nwx = W >> 3, nW = (H >> 3) * nwx;
  const long P = (long)H * W;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int gw = (blk / heads) * 4 + wave, h = blk % heads;
  if (gw * heads >= n_tasks) return;   // whole wave exits together; only wave-level synchronisation below
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
  float* __restrict__ vs = Vs[wave];
  float* __restrict__ ts = Ts[wave];

  // ---- operand fragments -----------------------------------------------------------------------------------
  float kf[2][WA_D / 2], qf[2][WA_D / 2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const float* __restrict__ kp = a.k.p + (long)img_k * a.k.bs + (long)(hk * WA_D + half) * P + pixel(wk, t2 * 32 + l31);
    const float* __restrict__ qp = a.q.p + (long)img * a.q.bs + (long)(h * WA_D + half) * P + pixel(w, t2 * 32 + l31);
#pragma unroll
    for (int s = 0; s < WA_D / 2; ++s) {
      kf[t2][s] = kp[(long)(2 * s) * P];
      qf[t2][s] = qp[(long)(2 * s) * P] * a.scale;
    }
  }
  {
    const float* __restrict__ vp = a.v.p + (long)img * a.v.bs + (long)(h * WA_D) * P + pixel(w, lane);
#pragma unroll 8
    for (int e = 0; e < WA_D; ++e) vs[lane * WA_LDV + e] = vp[(long)e * P];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {   // 225 table entries: four unpredicated loads per lane, in flight together
      const int i = lane + 64 * t4;
      const float tv = a.table[min(i, 224) * heads + h];
      if (i < 225) ts[i] = tv;
    }
  }
  // Swin shift-mask region of token `lane` on the shifted grid
  int rid_tok = 0;
  if (a.shift) {
    const int wy = w / nwx, wx = w - wy * nwx;
    const int sy = wy * 8 + (lane >> 3), sx = wx * 8 + (lane & 7);
    const int rh = sy < H - 8 ? 0 : (sy < H - a.shift ? 1 : 2);
    const int rw = sx < W - 8 ? 0 : (sx < W - a.shift ? 1 : 2);
    rid_tok = 3 * rh + rw;
  }

  float w_soft = a.w_softmax, w_relu = a.w_relu2;
  if (a.w_dev) {   // exp(w) / sum(exp(w)) as ast_arch.py:309-311 writes it
    const float e0 = expf(a.w_dev[0]), e1 = expf(a.w_dev[1]);
    w_soft = e0 / (e0 + e1);
    w_relu = e1 / (e0 + e1);
  }
  // ---- S^T = K Q^T ---------------------------------------------------------------------------------------------
  f32x16 S[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[mt][nt][r] = 0.f;
#pragma unroll
  for (int s = 0; s < WA_D / 2; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        S[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[mt][s], qf[nt][s], S[mt][nt], 0, 0, 0);

  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // ---- bias, mask, the two branches: per query column ----------------------------------------------------------
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 32 + l31;
    const int ny = n >> 3, nx = n & 7;
    const int rid_n = __shfl(rid_tok, n);
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float sc = S[mt][nt][r] + ts[(ny - (m >> 3) + 7) * 15 + (nx - (m & 7) + 7)];
        if (a.shift) {
          const int rid_m = __shfl(rid_tok, m);
          if (rid_m != rid_n) sc += -100.0f;
        }
        S[mt][nt][r] = sc;
        mx = fmaxf(mx, sc);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // each exponential once: kept in registers between the row sum and the mix
    float ex[2][16];
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ex[mt][r] = fast_exp(S[mt][nt][r] - mx);
        sum += ex[mt][r];
      }
    sum += __shfl_xor(sum, 32);
    const float inv_soft = w_soft / sum;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rl = fmaxf(S[mt][nt][r], 0.f);
        S[mt][nt][r] = fmaf(ex[mt][r], inv_soft, (rl * rl) * w_relu);
      }
  }

  // ---- O^T = V^T P^T -------------------------------------------------------------------------------------------
  f32x16 O[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[nt][r] = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float av = vs[m * WA_LDV + l31];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) O[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, S[mt][nt][r], O[nt], 0, 0, 0);
    }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float* __restrict__ op = a.out.p + (long)img * a.out.bs + (long)(h * WA_D + 4 * half) * P + pixel(w, nt * 32 + l31);
#pragma unroll
    for (int r = 0; r < 16; ++r) op[(long)((r & 3) + 8 * (r >> 2)) * P] = O[nt][r];
  }
}

void launch_window_attention(const WinAttnArgs& a, hipStream_t s) {
  JIRF_REQUIRE(a.B > 0 && a.heads > 0 && a.C == a.heads * WA_D, JIRF_EINVAL,
               "window attention: C=%d heads=%d (only head_dim 32 is built)", a.C, a.heads);
  JIRF_REQUIRE(a.H % 8 == 0 && a.W % 8 == 0 && a.H >= 8 && a.W >= 8, JIRF_EINVAL,
               "window attention: %dx%d is not a multiple of the 8x8 window", a.H, a.W);
  JIRF_REQUIRE(a.shift == 0 || a.shift == 4, JIRF_EINVAL, "window attention: shift %d", a.shift);
  const int n_tasks = a.B * (a.H / 8) * (a.W / 8) * a.heads;
  const double px = (double)a.B * a.H * a.W;
  Scope sc(intern_label("win_attn C%d H%d W%d", a.C, a.H, a.W), 4.0 * 64 * a.C * px, 16.0 * a.C * px, s);
  static const bool valu = std::getenv("JIRF_WINATTN_VALU") != nullptr;
  JIRF_REQUIRE(!(valu && a.w_dev), JIRF_EINVAL, "window attention: the vector-ALU form takes host mixing weights only");
  if (valu)
    hipLaunchKernelGGL(win_attn_kernel, dim3((n_tasks + 3) / 4), dim3(256), 0, s, a, n_tasks);
  else
    hipLaunchKernelGGL(win_attn_mfma_kernel, dim3((n_tasks / a.heads + 3) / 4 * a.heads), dim3(256), 0, s, a, n_tasks);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace jirf
