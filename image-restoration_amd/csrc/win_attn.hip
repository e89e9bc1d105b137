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

constexpr int WA_D = 32;       // head dim
constexpr int WA_N = 64;       // tokens per window
constexpr int WA_LD = 36;      // LDS row stride (floats): 16-byte aligned rows

__global__ __launch_bounds__(256, 3) void win_attn_kernel(const WinAttnArgs a, int n_tasks) {
  __shared__ __attribute__((aligned(16))) float KVs[4][WA_N * WA_LD];   // K during the scores, then V
  __shared__ float Ts[4][232];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + wave;
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
    for (int i = lane; i < 225; i += 64) ts[i] = a.table[i * heads + h];
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
  for (int m = 0; m < WA_N; ++m) sum += expf(sc[m] - mx);
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
    const float p = (expf(sc[m] - mx) * inv) * a.w_softmax + (r * r) * a.w_relu2;
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

void launch_window_attention(const WinAttnArgs& a, hipStream_t s) {
  JIRF_REQUIRE(a.B > 0 && a.heads > 0 && a.C == a.heads * WA_D, JIRF_EINVAL,
               "window attention: C=%d heads=%d (only head_dim 32 is built)", a.C, a.heads);
  JIRF_REQUIRE(a.H % 8 == 0 && a.W % 8 == 0 && a.H >= 8 && a.W >= 8, JIRF_EINVAL,
               "window attention: %dx%d is not a multiple of the 8x8 window", a.H, a.W);
  JIRF_REQUIRE(a.shift == 0 || a.shift == 4, JIRF_EINVAL, "window attention: shift %d", a.shift);
  const int n_tasks = a.B * (a.H / 8) * (a.W / 8) * a.heads;
  const double px = (double)a.B * a.H * a.W;
  Scope sc(intern_label("win_attn C%d H%d W%d", a.C, a.H, a.W), 4.0 * 64 * a.C * px, 16.0 * a.C * px, s);
  hipLaunchKernelGGL(win_attn_kernel, dim3((n_tasks + 3) / 4), dim3(256), 0, s, a, n_tasks);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace jirf
