// fp32 MFMA GEMM family of libjirf_hip (gfx950).
//
// One kernel template covers every dense contraction of the forward passes:
//   * 1x1 convolutions / Conv1d(k=1) / Linear:      Y[o][p] = sum_k W[o][k] X[k][p]
//   * attention * V with a gathered channel order (per-image "weight" = attention matrix)
//   * dense 3x3 convolutions as an implicit GEMM over k = tap * Cin + c
// Activations are planar ([channel][pixel], pixel contiguous), so the pixel axis sits on the
// MFMA column (lane) index: loads and stores are 128-byte row segments.  Arithmetic is
// v_mfma_f32_32x32x2_f32 (fp32 fma chains, k-ordered) in every kernel of this file but gemm_t3_kernel, which,
// like the kernels of gemm_split.hip, forms each fp32 product from exact bf16 partial products on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (on by default; launch_gemm picks per shape).
//
// Tile: a workgroup of WAVES_M x WAVES_N waves, each wave owning TM x TN accumulator tiles of
// 32 x 32; K stepped by 16 through a double-buffered LDS image [k][m] / [k][n]; operands are
// staged through registers so the loader can mask, gather, normalise (LayerNorm folded into the
// load) or shift (3x3 taps).  The 96-row tile exists because every channel count of the
// Restormer-family nets is 48 * 2^k: 96, 192, 288, 384, 576, 1152 are all multiples of 96.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "split_bf16.h"

namespace jirf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf(float v) { return gelu_fast(v); }

// Epilogue of one 32 x 32 accumulator tile: rows row0 + (r & 3) + 8 * (r >> 2) (row0 already holds
// the lane half's + 4), column `col` (< P).  Everything the epilogue reads is gathered first so
// the 16 loads are in flight together — output and residual may be the same buffer, which keeps
// the compiler from hoisting them itself — then combined and stored.
// MODE ST_FULL: the caller has checked (wave-uniformly) that all 32 rows are < M and all 32 columns
// < P, so nothing is predicated.  A predicated load makes the compiler branch around it and drain
// the memory counter at each join, which serialises the stores behind one another; for partial
// tiles ST_CLAMP therefore loads from clamped (always valid) addresses and predicates only the
// stores, at the cost of ~16 more registers; ST_PRED is the compact predicated form for kernels
// whose occupancy those registers would lower (partial tiles are the minority there).
enum { ST_FULL = 0, ST_CLAMP = 1, ST_PRED = 2 };
// Addresses are formed as (wave-uniform row pointer) + (32-bit lane offset): the row pointers live
// in scalar registers and one vector register holds the lane part for all 16 rows, instead of a
// 64-bit vector address per row.  rbase (multiple of 32, wave-uniform) is the tile's first row;
// lane half h owns rows rbase + (r & 3) + 8 * (r >> 2) + 4 h.
template <int EPI, int MODE>
__device__ __forceinline__ void store_tile(const GemmArgs& a, int b, const f32x16& acc, int rbase,
                                           int half, int col_in, const float* bias_r) {
  constexpr bool FULL = MODE == ST_FULL;
  const int M = a.M, P = a.P;
  const bool col_ok = FULL || col_in < P;
  if (MODE == ST_PRED && !col_ok) return;
  const unsigned voff = (unsigned)(4 * half) * (unsigned)P + (unsigned)col_in;
  auto urow = [&](int r) { return rbase + (r & 3) + 8 * (r >> 2); };
  auto lane_ok = [&](int r) { return FULL || (col_ok && urow(r) + 4 * half < M); };
  auto row_ptr = [&](auto* base, int r) { return base + (long)(FULL ? urow(r) : min(urow(r), M - 1)) * P; };
  float* __restrict__ yb = a.y.p + (long)b * a.y.bs;
  float aux[16];
  if constexpr (EPI == EP_RESID || EPI == EP_GELU_MUL) {
    const float* __restrict__ rp = a.r.p + (long)b * a.r.bs;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (MODE == ST_PRED) aux[r] = lane_ok(r) ? row_ptr(rp, r)[voff] : 0.f;
      else aux[r] = row_ptr(rp, r)[lane_ok(r) ? voff : 0u];
    }
  }
  if constexpr (EPI == EP_AV_INTRA) {
    const int* __restrict__ perm = a.kidx + (long)(b / a.imgs_per_group) * M;
    const int col = MODE == ST_CLAMP ? min(col_in, P - 1) : col_in;
    const float* __restrict__ qp = a.q.p + (long)b * a.q.bs + col;
    const float* __restrict__ kp = a.k.p + (long)b * a.k.bs + col;
    int src[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = urow(r) + 4 * half;
      if constexpr (MODE == ST_PRED) src[r] = row < M ? perm[row] : 0;
      else src[r] = perm[FULL ? row : min(row, M - 1)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      aux[r] = qp[(long)src[r] * P] * a.rq[(long)b * M + src[r]] +
               kp[(long)src[r] * P] * a.rk[(long)b * M + src[r]];
  }
  int py = 0, px = 0;
  if constexpr (EPI == EP_UNSHUFFLE || EPI == EP_SHUFFLE) {
    py = col_in / a.W;
    px = col_in - py * a.W;
  }
  float* __restrict__ y2b = nullptr;
  if constexpr (EPI == EP_AV_INTRA) y2b = a.y2.p + (long)b * a.y2.bs;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (!lane_ok(r)) continue;
    const float v = acc[r] + bias_r[r];
    if constexpr (EPI == EP_STORE) {
      row_ptr(yb, r)[voff] = v;
    } else if constexpr (EPI == EP_RESID) {
      row_ptr(yb, r)[voff] = v + aux[r];
    } else if constexpr (EPI == EP_GELU_MUL) {
      row_ptr(yb, r)[voff] = gelu_erf(v) * aux[r];
    } else if constexpr (EPI == EP_GELU) {
      row_ptr(yb, r)[voff] = gelu_erf(v);
    } else if constexpr (EPI == EP_LEAKY) {
      row_ptr(yb, r)[voff] = v > 0.f ? v : 0.01f * v;
    } else if constexpr (EPI == EP_AV_INTRA) {
      row_ptr(yb, r)[voff] = v;
      row_ptr(y2b, r)[voff] = v + aux[r];
    } else if constexpr (EPI == EP_UNSHUFFLE) {
      const int row = urow(r) + 4 * half;
      const int oc = 4 * row + 2 * (py & 1) + (px & 1);
      yb[(long)oc * (P >> 2) + (long)(py >> 1) * (a.W >> 1) + (px >> 1)] = v;
    } else if constexpr (EPI == EP_SHUFFLE) {
      const int row = urow(r) + 4 * half;
      const int oc = row >> 2;
      const int oy = 2 * py + ((row >> 1) & 1), ox = 2 * px + (row & 1);
      yb[(long)oc * (4L * P) + (long)oy * (2 * a.W) + ox] = v;
    }
  }
}

// Epilogue of TNW accumulator tiles whose columns are interleaved: lane l31 owns the TNW adjacent pixels
// col0 .. col0 + TNW - 1 (tile j holds pixel col0 + j), so one row of the lane's results is one 8- or
// 16-byte store and everything the epilogue reads comes in the same width.  A wave then issues 16 stores
// per group of tiles instead of 16 per tile: a wave may have 64 memory operations outstanding, and with
// one store per element the epilogue of a 32 x 128 block ran at the rate that limit and the write latency
// allow (~4 us), during which the wave issues no MFMA.  FULL: all 32 rows < M and every column < P;
// otherwise rows and column groups are predicated (loads from clamped, always valid addresses).
// P % TNW == 0 and 4 * TNW byte alignment of every row are the caller's responsibility.
template <int EPI, int TNW, bool FULL>
__device__ __forceinline__ void store_rows_v(const GemmArgs& a, int b, const f32x16* acc, int rbase, int half,
                                             int col0, const float* bias_r) {
  typedef float VT __attribute__((ext_vector_type(TNW)));
  const int M = a.M, P = a.P;
  const bool col_ok = FULL || col0 < P;
  const unsigned voff = (unsigned)(4 * half) * (unsigned)P + (unsigned)col0;
  auto urow = [&](int r) { return rbase + (r & 3) + 8 * (r >> 2); };
  auto lane_ok = [&](int r) { return FULL || (col_ok && urow(r) + 4 * half < M); };
#ifdef JIRF_XS_PROBE
  // ablation: every 32-row block stores over the tensor's first 32 rows (same instructions, cache-resident lines)
  auto row_ptr = [&](auto* base, int r) { return base + (long)((a.k_split & 8) ? (urow(r) & 31) : FULL ? urow(r) : min(urow(r), M - 1)) * P; };
#else
  auto row_ptr = [&](auto* base, int r) { return base + (long)(FULL ? urow(r) : min(urow(r), M - 1)) * P; };
#endif
  float* __restrict__ yb = a.y.p + (long)b * a.y.bs;
  if constexpr (EPI == EP_SHUFFLE) {   // scattered by construction: element stores
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int col = col0 + j;
      const int py = col / a.W, px = col - py * a.W;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!lane_ok(r)) continue;
        const int row = urow(r) + 4 * half;
        const int oy = 2 * py + ((row >> 1) & 1), ox = 2 * px + (row & 1);
        yb[(long)(row >> 2) * (4L * P) + (long)oy * (2 * a.W) + ox] = acc[j][r] + bias_r[r];
      }
    }
    return;
  }
  // rows go in chunks of RC: everything a chunk reads is requested before its first store, and the
  // operands of one chunk (not of all 16 rows) are what the epilogue adds to the register count
  constexpr int RC = (TNW >= 4 || EPI == EP_AV_INTRA) ? 4 : 8;
  float* __restrict__ y2b = nullptr;
  if constexpr (EPI == EP_AV_INTRA) y2b = a.y2.p + (long)b * a.y2.bs;
#pragma unroll
  for (int c0 = 0; c0 < 16; c0 += RC) {
    VT aux[RC];
    if constexpr (EPI == EP_RESID || EPI == EP_GELU_MUL) {
      const float* __restrict__ rp = a.r.p + (long)b * a.r.bs;
#pragma unroll
      for (int i = 0; i < RC; ++i)
        aux[i] = *reinterpret_cast<const VT*>(row_ptr(rp, c0 + i) + (lane_ok(c0 + i) ? voff : 0u));
    }
    if constexpr (EPI == EP_AV_INTRA) {
      const int* __restrict__ perm = a.kidx + (long)(b / a.imgs_per_group) * M;
      const int col = FULL ? col0 : (col_ok ? col0 : 0);
      const float* __restrict__ qp = a.q.p + (long)b * a.q.bs + col;
      const float* __restrict__ kp = a.k.p + (long)b * a.k.bs + col;
      int src[RC];
#pragma unroll
      for (int i = 0; i < RC; ++i) {
        const int row = urow(c0 + i) + 4 * half;
        src[i] = perm[FULL ? row : min(row, M - 1)];
      }
      VT qv[RC], kv[RC];
#pragma unroll
      for (int i = 0; i < RC; ++i) {
        qv[i] = *reinterpret_cast<const VT*>(qp + (long)src[i] * P);
        kv[i] = *reinterpret_cast<const VT*>(kp + (long)src[i] * P);
      }
#pragma unroll
      for (int i = 0; i < RC; ++i) aux[i] = qv[i] * a.rq[(long)b * M + src[i]] + kv[i] * a.rk[(long)b * M + src[i]];
    }
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      const int r = c0 + i;
      if (!lane_ok(r)) continue;
      VT v;
#pragma unroll
      for (int j = 0; j < TNW; ++j) v[j] = acc[j][r] + bias_r[r];
      if constexpr (EPI == EP_RESID) {
        v += aux[i];
      } else if constexpr (EPI == EP_GELU_MUL) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) v[j] = gelu_erf(v[j]) * aux[i][j];
      } else if constexpr (EPI == EP_GELU) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) v[j] = gelu_erf(v[j]);
      } else if constexpr (EPI == EP_LEAKY) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) v[j] = v[j] > 0.f ? v[j] : 0.01f * v[j];
      }
      *reinterpret_cast<VT*>(row_ptr(yb, r) + voff) = v;
      if constexpr (EPI == EP_AV_INTRA) *reinterpret_cast<VT*>(row_ptr(y2b, r) + voff) = v + aux[i];
    }
  }
}

template <int MODE>
__device__ __forceinline__ void load_bias16(const GemmArgs& a_in, int b, int rbase, int half, float* bias_r) {
  struct { const float* bias; int M; } a{a_in.bias ? a_in.bias + (long)b * a_in.bias_bs : nullptr, a_in.M};
  if (a.bias) {   // kernel argument: a scalar branch around the 16 loads
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * half;
      if constexpr (MODE == ST_PRED) bias_r[r] = row < a.M ? a.bias[row] : 0.f;
      else bias_r[r] = a.bias[MODE == ST_FULL ? row : min(row, a.M - 1)];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int LOADER, bool VEC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_kernel(const GemmArgs a_in,
                                                                        const int tiles_m,
                                                                        const int tiles_n,
                                                                        const int kt_per_split) {
  constexpr int BK = GEMM_BK;
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int WM = TM * 32, WN = TN * 32;
  constexpr int BM = WAVES_M * WM, BN = WAVES_N * WN;
  constexpr int XSLOTS = BK * BN / 4, WSLOTS = BK * BM / 4;
  constexpr int XS = (XSLOTS + NT - 1) / NT;  // float4 slots per thread, activation tile
  constexpr int WS = (WSLOTS + NT - 1) / NT;  // float4 slots per thread, weight tile
  constexpr bool X_FULL = XSLOTS % NT == 0;
  constexpr bool W_FULL = WSLOTS % NT == 0;

  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, half = lane >> 5;

  // split-K (gridDim.y > 1): this workgroup reduces the chunks [kt0, kt1) and leaves its raw sums in slab
  // blockIdx.y of the workspace; bias and epilogue belong to splitk_reduce_kernel
  GemmArgs a = a_in;
  const bool split = gridDim.y > 1;
  if (split) {
    a.y = TV{a_in.splitk_ws + (long)blockIdx.y * a_in.B * a_in.M * a_in.P, (long)a_in.M * a_in.P};
    a.bias = nullptr;
  }
  const int per_img = tiles_m * tiles_n;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int b = id / per_img;
  id -= b * per_img;
  const int tile_n = id / tiles_m;
  const int tile_m = id - tile_n * tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const float* __restrict__ wt = a.wt + (long)b * a.wt_bs;
  const float* __restrict__ xb = a.x.p + (long)b * a.x.bs;
  const int K = a.K, P = a.P;
  const int kt0 = (int)blockIdx.y * kt_per_split;
  const int nk = min((K + BK - 1) / BK, kt0 + kt_per_split);

  // ---- per-slot loader state ----------------------------------------------------------------
  int x_row[XS], x_col[XS];
  bool x_on[XS];
#pragma unroll
  for (int i = 0; i < XS; ++i) {
    const int s = tid + i * NT;
    x_row[i] = s / (BN / 4);
    x_col[i] = (s - x_row[i] * (BN / 4)) * 4;
    x_on[i] = X_FULL || s < XSLOTS;
  }
  float4 ln_mu[XS], ln_rs[XS];
  int cv_yx[XS][4];
  if constexpr (LOADER == LD_LN) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int p = n0 + x_col[i];
      float mu[4], rs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = x_on[i] && p + j < P;
        mu[j] = ok ? a.mu[(long)b * P + p + j] : 0.f;
        rs[j] = ok ? a.rstd[(long)b * P + p + j] : 0.f;
      }
      ln_mu[i] = make_float4(mu[0], mu[1], mu[2], mu[3]);
      ln_rs[i] = make_float4(rs[0], rs[1], rs[2], rs[3]);
    }
  }
  if constexpr (LOADER == LD_CONV3) {
#pragma unroll
    for (int i = 0; i < XS; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = n0 + x_col[i] + j;
        const int yy = p / a.W;
        // top-left input coordinate of the output pixel, biased by 8 so it packs unsigned
        cv_yx[i][j] = p < P ? (((yy * a.stride - a.pad + 8) << 16) | ((p - yy * a.W) * a.stride - a.pad + 8)) : -1;
      }
  }
  const int* __restrict__ kidx = nullptr;
  if constexpr (LOADER == LD_GATHER) kidx = a.kidx + (long)(b / a.imgs_per_group) * K;

  auto load_x = [&](int kt, int i) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k = kt * BK + x_row[i];
    if (!x_on[i] || k >= K) return v;
    const int p = n0 + x_col[i];
    if constexpr (LOADER == LD_CONV3) {
      const int tap = k / a.Cin;
      const int c = k - tap * a.Cin;
      const int ky = tap / a.ks;
      const int dy = ky - 8, dx = tap - ky * a.ks - 8;
      const float* __restrict__ plane = xb + (long)c * a.Hin * a.Win;
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yx = cv_yx[i][j];
        const int yy = (yx >> 16) + dy, xx = (yx & 0xffff) + dx;
        const bool ok = yx >= 0 && yy >= 0 && yy < a.Hin && xx >= 0 && xx < a.Win;
        r[j] = ok ? plane[yy * a.Win + xx] : 0.f;
      }
      return make_float4(r[0], r[1], r[2], r[3]);
    } else {
      int row = k;
      if constexpr (LOADER == LD_GATHER) row = kidx[k];
      const float* __restrict__ src = xb + (long)row * P + p;
      if constexpr (LOADER == LD_PLAIN) {
        if (k < a.k_split) src = a.x2.p + (long)b * a.x2.bs + (long)row * P + p;
      }
      if constexpr (VEC) {
        if (p < P) v = *reinterpret_cast<const float4*>(src);
      } else {
        if (p < P) v.x = src[0];
        if (p + 1 < P) v.y = src[1];
        if (p + 2 < P) v.z = src[2];
        if (p + 3 < P) v.w = src[3];
      }
      if constexpr (LOADER == LD_LN) {
        const float g = a.lnw ? a.lnw[k] : 1.f;
        const float be = a.lnb ? a.lnb[k] : 0.f;
        const float4 mu = ln_mu[i], rs = ln_rs[i];
        if (a.ln_center) {
          v.x = (v.x - mu.x) * rs.x * g + be;
          v.y = (v.y - mu.y) * rs.y * g + be;
          v.z = (v.z - mu.z) * rs.z * g + be;
          v.w = (v.w - mu.w) * rs.w * g + be;
        } else {
          v.x = v.x * rs.x * g;
          v.y = v.y * rs.y * g;
          v.z = v.z * rs.z * g;
          v.w = v.w * rs.w * g;
        }
        // columns past the image must stay zero (the bias term would otherwise leak in)
        if (p >= P) v.x = 0.f;
        if (p + 1 >= P) v.y = 0.f;
        if (p + 2 >= P) v.z = 0.f;
        if (p + 3 >= P) v.w = 0.f;
      }
      return v;
    }
  };
  auto load_w = [&](int kt, int i) -> float4 {
    const int s = tid + i * NT;
    if (!W_FULL && s >= WSLOTS) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int row = s / (BM / 4);
    const int col = m0 + (s - row * (BM / 4)) * 4;
    if (col >= a.Mpad) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(wt + (long)(kt * BK + row) * a.Mpad + col);
  };
  auto store_tiles = [&](int buf, const float4* xr, const float4* wr) {
#pragma unroll
    for (int i = 0; i < XS; ++i)
      if (x_on[i]) *reinterpret_cast<float4*>(&Bs[buf][x_row[i] * BN + x_col[i]]) = xr[i];
#pragma unroll
    for (int i = 0; i < WS; ++i) {
      const int s = tid + i * NT;
      if (W_FULL || s < WSLOTS) *reinterpret_cast<float4*>(&As[buf][s * 4]) = wr[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 xr[XS], wr[WS];
#pragma unroll
  for (int i = 0; i < XS; ++i) xr[i] = load_x(kt0, i);
#pragma unroll
  for (int i = 0; i < WS; ++i) wr[i] = load_w(kt0, i);
  store_tiles(kt0 & 1, xr, wr);
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < XS; ++i) xr[i] = load_x(kt + 1, i);
#pragma unroll
      for (int i = 0; i < WS; ++i) wr[i] = load_w(kt + 1, i);
    }
    __builtin_amdgcn_sched_barrier(0);   // the next chunk's fetches stay ahead of this chunk's MFMAs
    const float* __restrict__ Ab = &As[buf][wm * WM + l31];
    const float* __restrict__ Bb = &Bs[buf][wn * WN + l31];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = Ab[(kk + half) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bb[(kk + half) * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1, xr, wr);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
  auto run_epilogue = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = m0 + wm * WM + i * 32;
      if (rbase >= a.M) continue;   // wave-uniform: a padding tile
      float bias_r[16];
      if (rbase + 32 <= a.M) {
        load_bias16<ST_FULL>(a, b, rbase, half, bias_r);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cbase = n0 + wn * WN + j * 32;
          if (cbase + 32 <= P) store_tile<EPI, ST_FULL>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
          else if (cbase < P) store_tile<EPI, ST_PRED>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
        }
      } else {
        load_bias16<ST_PRED>(a, b, rbase, half, bias_r);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int cbase = n0 + wn * WN + j * 32;
          if (cbase < P) store_tile<EPI, ST_PRED>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
        }
      }
    }
  };
  // only the loader/epilogue pairs the networks use are instantiated
  const int epi = a.epi;
  if (split) {
    run_epilogue(std::integral_constant<int, EP_STORE>{});
    return;
  }
  if constexpr (LOADER == LD_GATHER) {
    run_epilogue(std::integral_constant<int, EP_AV_INTRA>{});
  } else if constexpr (LOADER == LD_LN) {
    run_epilogue(std::integral_constant<int, EP_STORE>{});
  } else if constexpr (LOADER == LD_CONV3) {
    if (epi == EP_UNSHUFFLE) run_epilogue(std::integral_constant<int, EP_UNSHUFFLE>{});
    else if (epi == EP_SHUFFLE) run_epilogue(std::integral_constant<int, EP_SHUFFLE>{});
    else if (epi == EP_RESID) run_epilogue(std::integral_constant<int, EP_RESID>{});
    else if (epi == EP_LEAKY) run_epilogue(std::integral_constant<int, EP_LEAKY>{});
    else run_epilogue(std::integral_constant<int, EP_STORE>{});
  } else {
    if (epi == EP_RESID) run_epilogue(std::integral_constant<int, EP_RESID>{});
    else if (epi == EP_GELU_MUL) run_epilogue(std::integral_constant<int, EP_GELU_MUL>{});
    else if (epi == EP_GELU) run_epilogue(std::integral_constant<int, EP_GELU>{});
    else if (epi == EP_SHUFFLE) run_epilogue(std::integral_constant<int, EP_SHUFFLE>{});
    else run_epilogue(std::integral_constant<int, EP_STORE>{});
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 form of the K-loop kernel (opt-in, see gemm_split.hip for the arithmetic): 128 x 128 tiles,
// four waves of 64 x 64, K stepped by 16.  The weights come pre-split from global memory ([plane][k / 8]
// [Mpad] 16-byte words); an activation element is split once per workgroup on its way to LDS: thread
// (k group g, pixel px) loads the eight channels 16 kt + 8 g .. + 7 of its pixel with one dword load each
// (128-byte row segments), normalises them if LayerNorm rides on the load, and writes the three bf16 planes
// as one 16-byte word each.  Both LDS images are [plane][k group][row or pixel] 16-byte words, so a wave's
// fragment read is 1 KiB contiguous: conflict-free.  24 MFMAs per wave between barriers.
// ------------------------------------------------------------------------------------------------
constexpr int T3_B = 128;

template <int LOADER>
__global__ __launch_bounds__(256, 3) void gemm_t3_kernel(const GemmArgs a, const int tiles_m, const int tiles_n) {
  __shared__ __attribute__((aligned(16))) u32x4 As[2][3 * 2 * T3_B];
  __shared__ __attribute__((aligned(16))) u32x4 Bs[2][3 * 2 * T3_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int g = wave >> 1;                       // the k group this thread stages (wave-uniform)
  const int srow = tid & (T3_B - 1);             // ... and its row (weights) / pixel (activations)
  const int per_img = tiles_m * tiles_n;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int b = id / per_img;
  id -= b * per_img;
  const int tile_n = id / tiles_m;
  const int tile_m = id - tile_n * tiles_m;
  const int m0 = tile_m * T3_B, n0 = tile_n * T3_B;
  const int K = a.K, P = a.P;
  const int nk = (K + 15) >> 4;

  const u32x4* __restrict__ W3 = reinterpret_cast<const u32x4*>(a.wt3) + min(m0 + srow, a.Mpad - 1);
  const int plane16 = (int)(a.wt3_plane / 8);
  const float* __restrict__ xb = a.x.p + (long)b * a.x.bs;
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)min((long)K * P * 4, 0x7fffffffL), 0x00020000);
  const float* __restrict__ xb2 = LOADER == LD_PLAIN && a.k_split > 0 ? a.x2.p + (long)b * a.x2.bs : xb;
  const __amdgpu_buffer_rsrc_t x2_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb2), 0, (int)min((long)K * P * 4, 0x7fffffffL), 0x00020000);
  const int pcl = min(n0 + srow, P - 1);         // columns past the image load a valid pixel and are never stored
  const int x_voff = pcl * 4;
  float ln_mu = 0.f, ln_rs = 1.f;
  if constexpr (LOADER == LD_LN) {
    ln_mu = a.ln_center ? a.mu[(long)b * P + pcl] : 0.f;
    ln_rs = a.rstd[(long)b * P + pcl];
  }

  // implicit k x k convolution: k = tap * Cin + c with Cin % 8 == 0, so the eight k of a fetch share one tap (a
  // scalar) and are eight adjacent input planes; the thread's input coordinate moves with the tap, the plane
  // advance is the scalar offset, and a tap that falls into the zero padding is an out-of-range offset: reads 0.
  int cv_y = 0, cv_x = 0;
  __amdgpu_buffer_rsrc_t cv_rsrc = x_rsrc;
  if constexpr (LOADER == LD_CONV3) {
    const int yy = pcl / a.W;
    cv_y = yy * a.stride - a.pad;
    cv_x = (pcl - yy * a.W) * a.stride - a.pad;
    cv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0,
                                                (int)min((long)a.Cin * a.Hin * a.Win * 4, 0x7fffffffL), 0x00020000);
  }

  u32x4 wa[3];
  float raw[8];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 3; ++i) wa[i] = W3[i * plane16 + (2 * kt + g) * a.Mpad];
    const int k0 = 16 * kt + 8 * g;              // wave-uniform: the row advance is a scalar offset
    if constexpr (LOADER == LD_CONV3) {
      const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
      const int ky = tap / a.ks, kx = tap - ky * a.ks;
      const int iy = cv_y + ky, ix = cv_x + kx;
      const bool ok = k0 < K && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      const int voff = ok ? (iy * a.Win + ix) * 4 : 0x7fffffff;
      const int plane_b = a.Hin * a.Win * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cv_rsrc, voff, (c0 + i) * plane_b, 0));
      return;
    }
    // rows past K read as zero through an out-of-range offset (their weights are zero padding; no branch)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + i;
      const int vo = k < K ? x_voff : 0x7fffffff;
      if (LOADER == LD_PLAIN && a.k_split > 0)
        raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(k < a.k_split ? x2_rsrc : x_rsrc, vo, min(k, K - 1) * P * 4, 0));
      else
        raw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, vo, min(k, K - 1) * P * 4, 0));
    }
  };
  auto stage = [&](int kt, int buf) {
    if constexpr (LOADER == LD_LN) {   // (the affine part is folded into the weights)
      const int k0 = 16 * kt + 8 * g;
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = k0 + i < K ? (raw[i] - ln_mu) * ln_rs : 0.f;
    }
    bf16x8 h, m, l;
    split8(raw, h, m, l);
    Bs[buf][0 * 2 * T3_B + tid] = __builtin_bit_cast(u32x4, h);
    Bs[buf][1 * 2 * T3_B + tid] = __builtin_bit_cast(u32x4, m);
    Bs[buf][2 * 2 * T3_B + tid] = __builtin_bit_cast(u32x4, l);
#pragma unroll
    for (int i = 0; i < 3; ++i) As[buf][i * 2 * T3_B + tid] = wa[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch(kt + 1);
    const bf16x8* __restrict__ Ab = reinterpret_cast<const bf16x8*>(&As[buf][half * T3_B + wm * 64 + l31]);
    const bf16x8* __restrict__ Bb = reinterpret_cast<const bf16x8*>(&Bs[buf][half * T3_B + wn * 64 + l31]);
    bf16x8 af[3][2], bfr[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[pl][i] = Ab[pl * 2 * T3_B + i * 32];
        bfr[pl][i] = Bb[pl * 2 * T3_B + i * 32];
      }
    // the six terms of a product, smallest first; consecutive MFMAs go to different accumulators
    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]][i], bfr[TB[t]][j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    __syncthreads();
  }

  auto run_epilogue = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = m0 + wm * 64 + i * 32;
      if (rbase >= a.M) continue;   // wave-uniform: a padding tile
      float bias_r[16];
      if (rbase + 32 <= a.M) {
        load_bias16<ST_FULL>(a, b, rbase, half, bias_r);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cbase = n0 + wn * 64 + j * 32;
          if (cbase + 32 <= P) store_tile<EPI, ST_FULL>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
          else if (cbase < P) store_tile<EPI, ST_PRED>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
        }
      } else {
        load_bias16<ST_PRED>(a, b, rbase, half, bias_r);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cbase = n0 + wn * 64 + j * 32;
          if (cbase < P) store_tile<EPI, ST_PRED>(a, b, acc[i][j], rbase, half, cbase + l31, bias_r);
        }
      }
    }
  };
  const int epi = a.epi;
  if constexpr (LOADER == LD_LN) {
    run_epilogue(std::integral_constant<int, EP_STORE>{});
  } else if constexpr (LOADER == LD_CONV3) {
    if (epi == EP_UNSHUFFLE) run_epilogue(std::integral_constant<int, EP_UNSHUFFLE>{});
    else if (epi == EP_SHUFFLE) run_epilogue(std::integral_constant<int, EP_SHUFFLE>{});
    else if (epi == EP_RESID) run_epilogue(std::integral_constant<int, EP_RESID>{});
    else if (epi == EP_LEAKY) run_epilogue(std::integral_constant<int, EP_LEAKY>{});
    else run_epilogue(std::integral_constant<int, EP_STORE>{});
  } else {
    if (epi == EP_RESID) run_epilogue(std::integral_constant<int, EP_RESID>{});
    else if (epi == EP_GELU_MUL) run_epilogue(std::integral_constant<int, EP_GELU_MUL>{});
    else if (epi == EP_GELU) run_epilogue(std::integral_constant<int, EP_GELU>{});
    else if (epi == EP_SHUFFLE) run_epilogue(std::integral_constant<int, EP_SHUFFLE>{});
    else run_epilogue(std::integral_constant<int, EP_STORE>{});
  }
}

// ------------------------------------------------------------------------------------------------
// Activation-stationary form for short reductions (K <= 256), which is every 1x1 conv of the full
// resolution levels.  The workgroup loads its whole [K][BN] activation tile into LDS once (gather,
// split source or LayerNorm applied on the way; LayerNorm statistics are computed from the tile
// itself, so no separate statistics pass exists), then each wave walks over 32-row output tiles:
// A fragments come straight from the K-major packed weights in L2 (128-byte rows), B fragments from
// LDS, and the stores of one output tile overlap the MFMAs of the next.  No barrier after the load.
// ------------------------------------------------------------------------------------------------
constexpr int XS_ABATCH = 8;  // k-steps of A fragments fetched ahead

// Diagnostic build only (-DJIRF_XS_PROBE, never the shipped library): shader-clock stamps at the phase
// boundaries of every wave of gemm_xs_kernel, read back through jirf_debug_xs_probe.
#ifdef JIRF_XS_PROBE
constexpr int XS_PROBE_SLOTS = 48, XS_PROBE_WGS = 8192;
__device__ unsigned long long g_xs_probe[XS_PROBE_WGS * 4 * XS_PROBE_SLOTS];
#define XS_STAMP(slot)                                                                                       \
  do {                                                                                                       \
    if (blockIdx.x < XS_PROBE_WGS && blockIdx.y == 0 && lane == 0 && (slot) < XS_PROBE_SLOTS)                 \
      g_xs_probe[((long)blockIdx.x * 4 + wave) * XS_PROBE_SLOTS + (slot)] = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define XS_STAMP(slot) do {} while (0)
#endif
#define XS_PROBE_SLOTS_LAST 47

template <int BN, int TNW, int LOADER, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_xs_kernel(const GemmArgs a, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float xs_smem[];
  constexpr int NG = BN / (32 * TNW);  // pixel groups a wave can own
  constexpr bool IL = VEC && TNW > 1;  // interleaved pixel columns (needs P % 4 == 0 and 16-byte aligned rows)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x - b * tiles_n) * BN;
  const int K = a.K, P = a.P, M = a.M;
  // K rounded up like the packed weights (zero rows there, zero rows here): the k loop then runs
  // in whole batches with no tail and no predicated load
  const int Kp = (K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  float* __restrict__ Xs = xs_smem;                  // [Kp][BN]
  float* __restrict__ st_mu = xs_smem + Kp * BN;     // [BN] (LD_LN)
  float* __restrict__ st_rs = st_mu + BN;
  const float* __restrict__ xb = a.x.p + (long)b * a.x.bs;
  const int* __restrict__ kidx = nullptr;
  if constexpr (LOADER == LD_GATHER) kidx = a.kidx + (long)(b / a.imgs_per_group) * K;

#ifdef JIRF_XS_PROBE
  if (a.k_split & 16) __builtin_amdgcn_s_setprio(3);
#endif
  XS_STAMP(0);
#ifdef JIRF_XS_PROBE
  if (blockIdx.x < XS_PROBE_WGS && blockIdx.y == 0 && lane == 0)
    g_xs_probe[((long)blockIdx.x * 4 + wave) * XS_PROBE_SLOTS + 1] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);   // HW_ID, XCC_ID
#endif
  // ---- load the tile --------------------------------------------------------------------------
  // In batches of XS_LB 16-byte loads per thread, all issued before the first LDS store: a loop of
  // load - wait - store keeps one load in flight per thread and leaves the tile load latency-bound.
  // The loads are unpredicated (clamped, always valid addresses; rejected slots are zeroed afterwards),
  // since a predicated load makes the compiler wait at the join.
  float* __restrict__ st_g = st_rs + BN;             // [Kp] LayerNorm weight (LD_LN)
  float* __restrict__ st_b = st_g + Kp;              // [Kp] LayerNorm bias
  // The output bias waits in LDS for the epilogues (rows past M as zeros): a vector-memory load there
  // would sit in the same in-order counter as the weight-fragment window and make the epilogue wait for it.
  float* __restrict__ st_bias = st_b + Kp;           // [ceil32(M)] when a.bias
  // ---- output tiles: bookkeeping and the first weight fragments -------------------------------------------
  const float* __restrict__ wt = a.wt + (long)b * a.wt_bs;
  const int n_mt = (M + 31) >> 5;
  // small grids split the output rows over gridDim.y workgroups (each reloads the pixel tile)
  const int mt_per = (n_mt + (int)gridDim.y - 1) / (int)gridDim.y;
  const int item_lo = min((int)blockIdx.y * mt_per, n_mt) * NG;
  const int n_items = min(((int)blockIdx.y + 1) * mt_per, n_mt) * NG;
  const int nsteps = Kp >> 1;
  const int epi = a.epi;
  // ---- the MFMA loop: a rolling window instead of batches ------------------------------------------
  // A wave alone on its SIMD must keep the matrix pipe fed (the workgroups sharing the SIMD spend most of
  // their life in load / LayerNorm / store phases).  So nothing waits at a batch boundary: the weight
  // fragment of k-step s + 8 is requested into the register that k-step s has just used, the LDS fragment
  // of k-step s + 2 likewise, and both windows run on into the wave's next item.  Weights are read through
  // a buffer descriptor: the row advances in the scalar offset, the lane part is one register per item, and
  // rows past Mpad do not exist in the packed weights, so the column is clamped (such rows are never stored).
  // The first window is requested here, before the tile is even loaded: the weights do not depend on the tile,
  // the loads are older than the tile's in the in-order counter, and the first item starts without an L2 round trip.
  typedef float BV __attribute__((ext_vector_type(TNW)));
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, Kp * a.Mpad * 4, 0x00020000);
  const int row2 = 2 * a.Mpad * 4;                      // bytes per k-step (two weight rows)
  auto a_voff = [&](int item) { return (half * a.Mpad + min((item / NG) * 32 + l31, a.Mpad - 1)) * 4; };
  float a_cur[XS_ABATCH];
  if (item_lo + wave < n_items) {
    const int v0 = a_voff(item_lo + wave);
#pragma unroll
    for (int i = 0; i < XS_ABATCH; ++i) a_cur[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(w_rsrc, v0, i * row2, 0));
  }
  // (requested now from clamped addresses, written to LDS just before the barrier: the wait for these loads
  // would otherwise stand in front of the tile's own loads)
  constexpr int BIAS_SLOTS = 8;                      // M <= 2048; larger M stages the rest in a plain loop
  const int bias_n = a.bias ? min(BIAS_SLOTS, (((M + 31) & ~31) + 255) >> 8) : 0;
  float bias_v[BIAS_SLOTS];
  if (a.bias) {
    const float* __restrict__ bsrc = a.bias + (long)b * a.bias_bs;
#pragma unroll
    for (int i = 0; i < BIAS_SLOTS; ++i)
      if (i < bias_n) bias_v[i] = bsrc[min(tid + 256 * i, M - 1)];
  }
  auto stage_bias = [&]() {
    if (!a.bias) return;
    const int mc = (M + 31) & ~31;
#pragma unroll
    for (int i = 0; i < BIAS_SLOTS; ++i)
      if (i < bias_n && tid + 256 * i < mc) st_bias[tid + 256 * i] = tid + 256 * i < M ? bias_v[i] : 0.f;
    const float* __restrict__ bsrc = a.bias + (long)b * a.bias_bs;
    for (int m = tid + 256 * BIAS_SLOTS; m < mc; m += 256) st_bias[m] = m < M ? bsrc[m] : 0.f;
  };
  if constexpr (LOADER == LD_LN && VEC) {
    // LayerNorm in registers.  A wave owns CW groups of four pixels; a lane holds every RW-th channel of
    // its group, so all channels of a pixel live in one wave: mean and variance (two passes over the
    // registers, as the reference's mean / var(unbiased=False)) are lane-exchange sums, and the tile
    // reaches LDS once, already normalised.  One barrier (before the waves read each other's columns).
    constexpr int CW = BN / 16;            // groups of four pixels per wave: 8 (BN = 128) or 4 (BN = 64)
    constexpr int RW = 64 / CW;            // channels one load instruction covers: 8 or 16
    constexpr int NSMAX = 16;              // register slots: Kp <= 16 RW is checked at launch
    const int cg = wave * CW + (lane % CW), r0 = lane / CW;
    const int ns = Kp / RW;                // Kp is a multiple of 16
    const int p = n0 + cg * 4;
    const bool col_ok = p < P;             // P % 4 == 0 here: the four pixels of a group stand or fall together
    const float* __restrict__ src = xb + (col_ok ? p : 0);
    // affine == false: the caller has folded the LayerNorm weight into the GEMM weights and its bias into
    // the output bias (pack_upload_ln_1x1), and the tile is only normalised here
    const bool affine = a.lnw != nullptr, has_b = a.lnb != nullptr;
    // Fast path (every LayerNorm GEMM of the four networks): K a multiple of 16 (no padding rows), affine folded
    // into the weights.  The slot count is a compile-time constant, nothing is predicated (columns past the image
    // are loaded from clamped addresses, normalised like the rest and never stored): about half of the general
    // path's vector instructions - they are issued while the other workgroups of the CU keep the matrix pipe
    // busy, which starves them (each costs > 100 cycles).
    auto ln_fast = [&](auto ns_tag) {
      constexpr int NS = decltype(ns_tag)::value;
      // One buffer descriptor per image; the lane part of the address is one register, the row advance of slot u
      // is a scalar offset: the twelve loads of a K = 96 tile cost three vector instructions of address arithmetic
      // instead of one 64-bit multiply-add chain each (vector instructions are what this phase is short of).
      typedef unsigned U4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t x_rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)min((long)K * P * 4, 0x7fffffffL), 0x00020000);
      const int voff = (r0 * P + (col_ok ? p : 0)) * 4;
      const int row_step = RW * P * 4;
      // the arithmetic on two-float vectors, so that it compiles to the packed fp32 instructions (v_pk_add_f32,
      // v_pk_mul_f32, v_pk_fma_f32: half the instruction count), and 1 / sqrt as v_rsq_f32 (1 ulp)
      typedef float F2 __attribute__((ext_vector_type(2)));
      F2 va[NS], vb[NS];                     // pixels (0, 1) and (2, 3) of the lane's group, slot u
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const U4 t = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, u * row_step, 0);
        va[u] = F2{__uint_as_float(t[0]), __uint_as_float(t[1])};
        vb[u] = F2{__uint_as_float(t[2]), __uint_as_float(t[3])};
      }
      XS_STAMP(40);
      // Sum over the lanes that hold the other channels of the same pixels (lane bits log2(CW) .. 5), on the
      // vector ALU only: rotations inside a row of 16 lanes (DPP) and the two row-swap instructions of gfx950,
      // v_permlane16_swap / v_permlane32_swap (with both operands the same value, the swapped pair adds up to
      // the butterfly sum).  The LDS crossbar (ds_bpermute) is busy with the B fragments of the MFMA waves.
      auto bits = [](float f) { return __builtin_bit_cast(unsigned, f); };
      auto flt = [](unsigned u) { return __builtin_bit_cast(float, u); };
      auto lane_sum1 = [&](float t) {
        if constexpr (CW == 4) t += flt(__builtin_amdgcn_update_dpp(0, bits(t), 0x124, 0xf, 0xf, false));   // row_ror:4
        t += flt(__builtin_amdgcn_update_dpp(0, bits(t), 0x128, 0xf, 0xf, false));                          // row_ror:8
        const auto s16 = __builtin_amdgcn_permlane16_swap(bits(t), bits(t), false, false);
        t = flt(s16[0]) + flt(s16[1]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(bits(t), bits(t), false, false);
        return flt(s32[0]) + flt(s32[1]);
      };
      auto lane_sum2 = [&](F2& t) { t = F2{lane_sum1(t[0]), lane_sum1(t[1])}; };
      F2 sa = va[0], sb = vb[0];
#pragma unroll
      for (int u = 1; u < NS; ++u) { sa += va[u]; sb += vb[u]; }
      XS_STAMP(2);
      lane_sum2(sa);
      lane_sum2(sb);
      XS_STAMP(42);
      const float inv_k = 1.0f / (float)K;
      const F2 mua = sa * inv_k, mub = sb * inv_k;
      F2 qa = F2{0.f, 0.f}, qb = F2{0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const F2 da = va[u] - mua, db = vb[u] - mub;
        qa = __builtin_elementwise_fma(da, da, qa);
        qb = __builtin_elementwise_fma(db, db, qb);
      }
      lane_sum2(qa);
      lane_sum2(qb);
      XS_STAMP(43);
      const F2 ea = qa * inv_k + 1e-5f, eb = qb * inv_k + 1e-5f;
      const F2 rsa = F2{__builtin_amdgcn_rsqf(ea[0]), __builtin_amdgcn_rsqf(ea[1])};
      const F2 rsb = F2{__builtin_amdgcn_rsqf(eb[0]), __builtin_amdgcn_rsqf(eb[1])};
      const F2 mca = a.ln_center ? mua : F2{0.f, 0.f}, mcb = a.ln_center ? mub : F2{0.f, 0.f};   // BiasFree does not centre
      float4* __restrict__ xs_w = reinterpret_cast<float4*>(&Xs[r0 * BN + cg * 4]);
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const F2 oa = (va[u] - mca) * rsa, ob = (vb[u] - mcb) * rsb;
        xs_w[u * (RW * BN / 4)] = make_float4(oa[0], oa[1], ob[0], ob[1]);
      }
    };
    bool fast = false;
    if (!affine && Kp == K && (long)K * P * 4 < 0x7fffffffL) {
      fast = true;
      switch (ns) {
        case 2: ln_fast(std::integral_constant<int, 2>{}); break;
        case 3: ln_fast(std::integral_constant<int, 3>{}); break;
        case 4: ln_fast(std::integral_constant<int, 4>{}); break;
        case 6: ln_fast(std::integral_constant<int, 6>{}); break;
        case 8: ln_fast(std::integral_constant<int, 8>{}); break;
        case 12: ln_fast(std::integral_constant<int, 12>{}); break;
        case 16: ln_fast(std::integral_constant<int, 16>{}); break;
        default: fast = false;
      }
    }
    if (!fast) {
    float4 v[NSMAX];
    float g[NSMAX], be[NSMAX];
#pragma unroll
    for (int h = 0; h < NSMAX / 2; ++h) {
      if (2 * h < ns) {                    // wave-uniform: whole pairs of loads are skipped, none is predicated
#pragma unroll
        for (int u = 2 * h; u < 2 * h + 2; ++u) {
          const int k = min(r0 + RW * u, K - 1);
#ifdef JIRF_XS_PROBE
          if (a.k_split & 1) v[u] = make_float4(0.01f * lane, 0.1f * u, 1.f, -0.3f * k);   // ablation: no tile load
          else
#endif
          v[u] = *reinterpret_cast<const float4*>(src + (long)k * P);
          if (affine) {
            g[u] = a.lnw[k];
            be[u] = has_b ? a.lnb[k] : 0.f;
          }
        }
      }
    }
    XS_STAMP(40);
    auto lane_sum = [&](float4& t) {
#pragma unroll
      for (int m = CW; m < 64; m <<= 1) {
        t.x += __shfl_xor(t.x, m);
        t.y += __shfl_xor(t.y, m);
        t.z += __shfl_xor(t.z, m);
        t.w += __shfl_xor(t.w, m);
      }
    };
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NSMAX; ++u) {
      if (u < ns) {
        if (r0 + RW * u >= K) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);   // a padding row (clamped load)
        sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w;
      }
    }
    XS_STAMP(2);
    lane_sum(sum);
    XS_STAMP(42);
    const float inv_k = 1.0f / (float)K;
    const float4 mu = make_float4(sum.x * inv_k, sum.y * inv_k, sum.z * inv_k, sum.w * inv_k);
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NSMAX; ++u) {
      if (u < ns && r0 + RW * u < K) {
        const float dx = v[u].x - mu.x, dy = v[u].y - mu.y, dz = v[u].z - mu.z, dw = v[u].w - mu.w;
        m2.x = fmaf(dx, dx, m2.x); m2.y = fmaf(dy, dy, m2.y); m2.z = fmaf(dz, dz, m2.z); m2.w = fmaf(dw, dw, m2.w);
      }
    }
    lane_sum(m2);
    XS_STAMP(43);
    const float4 rs = make_float4(1.0f / sqrtf(m2.x * inv_k + 1e-5f), 1.0f / sqrtf(m2.y * inv_k + 1e-5f),
                                  1.0f / sqrtf(m2.z * inv_k + 1e-5f), 1.0f / sqrtf(m2.w * inv_k + 1e-5f));
    const float4 mc = a.ln_center ? mu : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NSMAX; ++u) {
      if (u < ns) {
        const int k = r0 + RW * u;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && col_ok) {             // rows past K and columns past the image stay zero
          o.x = (v[u].x - mc.x) * rs.x;
          o.y = (v[u].y - mc.y) * rs.y;
          o.z = (v[u].z - mc.z) * rs.z;
          o.w = (v[u].w - mc.w) * rs.w;
          if (affine) {
            o.x = o.x * g[u] + be[u];
            o.y = o.y * g[u] + be[u];
            o.z = o.z * g[u] + be[u];
            o.w = o.w * g[u] + be[u];
          }
        }
        *reinterpret_cast<float4*>(&Xs[k * BN + cg * 4]) = o;
      }
    }
    }
    XS_STAMP(44);
    stage_bias();
    __syncthreads();
  } else {
    if constexpr (LOADER == LD_LN) {
      for (int k = tid; k < K; k += 256) {
        st_g[k] = a.lnw ? a.lnw[k] : 1.f;
        st_b[k] = a.lnb ? a.lnb[k] : 0.f;
      }
    }
    if constexpr (VEC) {
      constexpr int XS_LB = 12;
      const int total = Kp * (BN / 4);
      const float* __restrict__ xb2 = LOADER == LD_PLAIN && a.k_split > 0 ? a.x2.p + (long)b * a.x2.bs : xb;
      for (int s0 = tid; s0 < total; s0 += 256 * XS_LB) {
        float4 v[XS_LB];
        bool ok[XS_LB];
  #pragma unroll
        for (int u = 0; u < XS_LB; ++u) {
          const int s = s0 + u * 256;
          const int k = s / (BN / 4), c4 = (s - k * (BN / 4)) * 4;
          const int p = n0 + c4;
          ok[u] = s < total && k < K && p < P;
          const int kc = min(k, K - 1);
          int row = kc;
          if constexpr (LOADER == LD_GATHER) row = kidx[kc];
          const float* __restrict__ base = (LOADER == LD_PLAIN && kc < a.k_split) ? xb2 : xb;
          v[u] = *reinterpret_cast<const float4*>(base + (long)row * P + (p < P ? p : 0));
        }
  #pragma unroll
        for (int u = 0; u < XS_LB; ++u) {
          const int s = s0 + u * 256;
          if (s < total) {
            const int k = s / (BN / 4), c4 = (s - k * (BN / 4)) * 4;
            *reinterpret_cast<float4*>(&Xs[k * BN + c4]) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    } else {
      for (int s = tid; s < Kp * (BN / 4); s += 256) {
        const int k = s / (BN / 4), c4 = (s - k * (BN / 4)) * 4;
        const int p = n0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && p < P) {
          int row = k;
          if constexpr (LOADER == LD_GATHER) row = kidx[k];
          const float* __restrict__ src = xb + (long)row * P + p;
          if constexpr (LOADER == LD_PLAIN) {
            if (k < a.k_split) src = a.x2.p + (long)b * a.x2.bs + (long)row * P + p;
          }
          v.x = src[0];
          if (p + 1 < P) v.y = src[1];
          if (p + 2 < P) v.z = src[2];
          if (p + 3 < P) v.w = src[3];
        }
        *reinterpret_cast<float4*>(&Xs[k * BN + c4]) = v;
      }
    }
    stage_bias();
    __syncthreads();
    XS_STAMP(2);
    if constexpr (LOADER == LD_LN) {
      // per-pixel mean and population variance over the K channels (two passes, as the reference's
      // mean / var(unbiased=False)); columns past the image stay zero
      if (tid < BN) {
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += Xs[k * BN + tid];
        const float mu = sum / (float)K;
        float var = 0.f;
        for (int k = 0; k < K; ++k) {
          const float d = Xs[k * BN + tid] - mu;
          var = fmaf(d, d, var);
        }
        st_mu[tid] = a.ln_center ? mu : 0.f;
        st_rs[tid] = n0 + tid < P ? 1.0f / sqrtf(var / (float)K + 1e-5f) : 0.f;
      }
      __syncthreads();
  #pragma unroll 4
      for (int s = tid; s < K * (BN / 4); s += 256) {
        const int k = s / (BN / 4), c4 = (s - k * (BN / 4)) * 4;
        const float g = st_g[k];
        const float be = st_b[k];
        float4 v = *reinterpret_cast<float4*>(&Xs[k * BN + c4]);
        const float4 mu = *reinterpret_cast<const float4*>(&st_mu[c4]);
        const float4 rs = *reinterpret_cast<const float4*>(&st_rs[c4]);
        const int p = n0 + c4;
        v.x = p < P ? (v.x - mu.x) * rs.x * g + be : 0.f;
        v.y = p + 1 < P ? (v.y - mu.y) * rs.y * g + be : 0.f;
        v.z = p + 2 < P ? (v.z - mu.z) * rs.z * g + be : 0.f;
        v.w = p + 3 < P ? (v.w - mu.w) * rs.w * g + be : 0.f;
        *reinterpret_cast<float4*>(&Xs[k * BN + c4]) = v;
      }
      __syncthreads();
    }
  }

  XS_STAMP(3);
  auto b_ptr = [&](int item) { return Xs + half * BN + (item % NG) * TNW * 32 + (IL ? TNW * l31 : l31); };
  auto load_b = [&](const float* __restrict__ q) {
    BV v;
    if constexpr (IL) {
      v = *reinterpret_cast<const BV*>(q);
    } else {
#pragma unroll
      for (int j = 0; j < TNW; ++j) v[j] = q[j * 32];
    }
    return v;
  };
  BV b_cur[2];
  if (item_lo + wave < n_items) {
    const float* __restrict__ b0 = b_ptr(item_lo + wave);
    b_cur[0] = load_b(b0);
    b_cur[1] = load_b(b0 + 2 * BN);
  }
  for (int item = item_lo + wave; item < n_items; item += 4) {
    const int mt = item / NG, ng = item - mt * NG;
    const int m0 = mt * 32;
    const int col_w = ng * TNW * 32;            // first pixel of this wave's group inside the tile
    f32x16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int item_n = item + 4 < n_items ? item + 4 : item;
    const int av = a_voff(item), av_n = a_voff(item_n);
    const float* __restrict__ bp = b_ptr(item);
    const float* __restrict__ bp_n = b_ptr(item_n);
#ifdef JIRF_XS_PROBE
    if (a.k_split & 48) __builtin_amdgcn_s_setprio(0);
#endif
    for (int s0 = 0; s0 < nsteps; s0 += XS_ABATCH) {
      const bool last = s0 + XS_ABATCH >= nsteps;
      const int a_v = last ? av_n : av;                               // window k-steps s0 + 8 ...
      const int a_s = last ? 0 : (s0 + XS_ABATCH) * row2;
      const float* __restrict__ b_c = bp + (2 * s0) * BN;
      const float* __restrict__ b_n = last ? bp_n : bp + (2 * (s0 + XS_ABATCH)) * BN;
#pragma unroll
      for (int i = 0; i < XS_ABATCH; ++i) {
#pragma unroll
        for (int j = 0; j < TNW; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[i & 1][j], acc[j], 0, 0, 0);
        a_cur[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(w_rsrc, a_v, a_s + i * row2, 0));
        b_cur[i & 1] = load_b(i + 2 < XS_ABATCH ? b_c + (2 * (i + 2)) * BN : b_n + (2 * (i + 2 - XS_ABATCH)) * BN);
        // one k-step = TNW MFMAs, one weight fetch, one LDS fetch, in that order
        __builtin_amdgcn_sched_group_barrier(0x8, TNW, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, IL ? 1 : TNW, 0);
      }
    }
#ifdef JIRF_XS_PROBE
    if (a.k_split & 32) __builtin_amdgcn_s_setprio(3);
#endif
    XS_STAMP(4 + 2 * ((item - item_lo) >> 2));
    float bias_r[16];
    auto store_all = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      constexpr int ST = FULL ? ST_FULL : ST_CLAMP;
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = st_bias[m0 + (r & 3) + 8 * (r >> 2) + 4 * half];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
      }
      if constexpr (IL) {
        const int col0 = n0 + col_w + TNW * l31;
        if constexpr (LOADER == LD_GATHER) {
          store_rows_v<EP_AV_INTRA, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
        } else if constexpr (LOADER == LD_LN) {
          store_rows_v<EP_STORE, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
        } else {
          if (epi == EP_RESID) store_rows_v<EP_RESID, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
          else if (epi == EP_GELU_MUL) store_rows_v<EP_GELU_MUL, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
          else if (epi == EP_GELU) store_rows_v<EP_GELU, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
          else if (epi == EP_SHUFFLE) store_rows_v<EP_SHUFFLE, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
          else store_rows_v<EP_STORE, TNW, FULL>(a, b, acc, m0, half, col0, bias_r);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int col = n0 + col_w + j * 32 + l31;
        if (!FULL && n0 + col_w + j * 32 >= P) continue;   // wave-uniform
        if constexpr (LOADER == LD_GATHER) {
          store_tile<EP_AV_INTRA, ST>(a, b, acc[j], m0, half, col, bias_r);
        } else if constexpr (LOADER == LD_LN) {
          store_tile<EP_STORE, ST>(a, b, acc[j], m0, half, col, bias_r);
        } else {
          if (epi == EP_RESID) store_tile<EP_RESID, ST>(a, b, acc[j], m0, half, col, bias_r);
          else if (epi == EP_GELU_MUL) store_tile<EP_GELU_MUL, ST>(a, b, acc[j], m0, half, col, bias_r);
          else if (epi == EP_GELU) store_tile<EP_GELU, ST>(a, b, acc[j], m0, half, col, bias_r);
          else if (epi == EP_SHUFFLE) store_tile<EP_SHUFFLE, ST>(a, b, acc[j], m0, half, col, bias_r);
          else store_tile<EP_STORE, ST>(a, b, acc[j], m0, half, col, bias_r);
        }
      }
    };
#ifdef JIRF_XS_PROBE
    if ((a.k_split & 4) && a.epi != 12345) {} else                                       // ablation: no stores
#endif
    if (m0 + 32 <= M && n0 + BN <= P) store_all(std::true_type{});
    else store_all(std::false_type{});
    XS_STAMP(5 + 2 * ((item - item_lo) >> 2));
  }
  XS_STAMP(XS_PROBE_SLOTS_LAST);
}

#ifdef JIRF_XS_PROBE
extern "C" int jirf_debug_xs_probe(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xs_probe), sizeof(unsigned long long) * XS_PROBE_WGS * 4 * XS_PROBE_SLOTS);
}
#endif

// ------------------------------------------------------------------------------------------------
// Weight-stationary form for short outputs (M <= 96: the project_out / gate / attention-tail GEMMs,
// whose activations are several times wider than their outputs, so they run at the HBM bound).
// The whole packed weight matrix sits in LDS ([Kp][MT * 32], loaded once per workgroup); every wave
// is its own pipeline with no barrier after that: it walks over groups of 64 pixels, streaming the
// activation rows straight from global memory into MFMA B fragments (a lane holds two adjacent
// pixels of rows k, k + 1: 256 contiguous bytes per row and load) through a window of WS_WIN k-steps
// that runs on into the wave's next pixel group, so ~WS_WIN loads per wave stay in flight against the
// HBM latency; A fragments come from LDS (one ds_read per 32-row tile and k-step, feeding 2 MFMAs).
// The activation tensor is read exactly once and nothing but the weights passes through LDS.
// ------------------------------------------------------------------------------------------------
constexpr int WS_WIN = 16;     // k-steps of activation fragments in flight per wave
constexpr int WS_WAVES = 8;

template <int MT, int EPI>
__global__ __launch_bounds__(64 * WS_WAVES, 1) void gemm_ws_kernel(const GemmArgs a, const int groups_per_img,
                                                                   const int total_groups) {
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  typedef float BV __attribute__((ext_vector_type(2)));
  typedef unsigned UV __attribute__((ext_vector_type(2)));
  constexpr int MW = MT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int K = a.K, P = a.P, M = a.M;
  const int Kp = (K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const int nsteps = Kp >> 1;
  float* __restrict__ Ws = ws_smem;            // [Kp][MW]
  float* __restrict__ st_bias = Ws + Kp * MW;  // [MW]
  // ---- weights (and bias) into LDS: rows of the K-major packed matrix, first MW columns ---------------
  for (int i = tid; i < Kp * (MW / 4); i += 64 * WS_WAVES) {
    const int k = i / (MW / 4), c4 = (i - k * (MW / 4)) * 4;
    *reinterpret_cast<float4*>(&Ws[k * MW + c4]) = *reinterpret_cast<const float4*>(a.wt + (long)k * a.Mpad + c4);
  }
  for (int m = tid; m < MW; m += 64 * WS_WAVES) st_bias[m] = (a.bias && m < M) ? a.bias[m] : 0.f;
  __syncthreads();

  // a wave's pixel groups: g = first, first + stride, ...  (64 pixels each, never across images)
  // With fewer groups than waves (a single 256 x 256 image has 1024) wave w of workgroup b starts at group
  // w * gridDim.x + b, so that every CU still gets its share, on one wave per SIMD.
  const int stride = (int)gridDim.x * WS_WAVES;
  int g = total_groups >= stride ? (int)blockIdx.x * WS_WAVES + wave : wave * (int)gridDim.x + (int)blockIdx.x;
  if (g >= total_groups) return;
  const int row2 = 2 * P * 4;                   // bytes per k-step (two activation rows)
  // one descriptor per image: rows past K read as zero (range check), pixels past P belong to discarded columns
  auto rsrc_of = [&](int grp) {
    const int b = grp / groups_per_img;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x.p + (long)b * a.x.bs), 0, K * P * 4, 0x00020000);
  };
  auto voff_of = [&](int grp) {
    const int b = grp / groups_per_img;
    return (half * P + (grp - b * groups_per_img) * 64 + 2 * l31) * 4;
  };
  auto load_b = [&](__amdgpu_buffer_rsrc_t rs, int voff, int s) {
    return __builtin_bit_cast(BV, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, s * row2, 0));
  };
  BV bw[WS_WIN];
  {
    const __amdgpu_buffer_rsrc_t r0 = rsrc_of(g);
    const int v0 = voff_of(g);
#pragma unroll
    for (int i = 0; i < WS_WIN; ++i) bw[i] = load_b(r0, v0, i);   // nsteps >= WS_WIN is checked at launch
  }
  const float* __restrict__ ap = Ws + half * MW + l31;
  while (g < total_groups) {
    const int gn = g + stride < total_groups ? g + stride : g;
    const __amdgpu_buffer_rsrc_t rs_c = rsrc_of(g), rs_n = rsrc_of(gn);
    const int vo_c = voff_of(g), vo_n = voff_of(gn);
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;
    float af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = ap[mt * 32];
    // Batches of WS_WIN k-steps: the activation fragments of the NEXT batch (of the wave's next pixel group
    // after the last batch) are requested up front, all WS_WIN of them, and are due only after this batch's
    // 2 MT WS_WIN MFMAs (~2.5 us of matrix-pipe time at MT = 3): the HBM latency hides behind them.
    for (int s0 = 0; s0 < nsteps; s0 += WS_WIN) {
      const bool last = s0 + WS_WIN >= nsteps;
      const __amdgpu_buffer_rsrc_t rs = last ? rs_n : rs_c;
      const int vo = last ? vo_n : vo_c;
      const int sb = last ? 0 : s0 + WS_WIN;
      BV bn[WS_WIN];
#pragma unroll
      for (int i = 0; i < WS_WIN; ++i) bn[i] = load_b(rs, vo, sb + i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WS_WIN; ++i) {
        // the A fragments of the next k-step are requested before this k-step's MFMAs (the group's last
        // k-step re-reads row 0: the same for every pixel group)
        float an[MT];
        const int sn = s0 + i + 1 < nsteps ? s0 + i + 1 : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) an[mt] = ap[(2 * sn) * MW + mt * 32];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt], bw[i][j], acc[mt][j], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = an[mt];
      }
#pragma unroll
      for (int i = 0; i < WS_WIN; ++i) bw[i] = bn[i];
    }
    // ---- epilogue -----------------------------------------------------------------------------------
    const int b = g / groups_per_img;
    const int col0 = (g - b * groups_per_img) * 64 + 2 * l31;
    const bool full = MT * 32 <= M && (g - b * groups_per_img) * 64 + 64 <= P;   // every tile of the group alike
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m0 = mt * 32;
      if (m0 >= M) continue;
      float bias_r[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[r] = st_bias[m0 + (r & 3) + 8 * (r >> 2) + 4 * half];
      if (full) store_rows_v<EPI, 2, true>(a, b, acc[mt], m0, half, col0, bias_r);
      else store_rows_v<EPI, 2, false>(a, b, acc[mt], m0, half, col0, bias_r);
    }
    g += stride;
  }
}

static bool try_launch_ws(const GemmArgs& a, int loader, bool vec, hipStream_t s) {
  static const bool off = std::getenv("JIRF_GEMM_NO_WS") != nullptr || std::getenv("JIRF_GEMM_TILE") != nullptr;
  if (off || loader != LD_PLAIN || !vec || a.M > 96 || a.wt_bs != 0 || a.bias_bs != 0) return false;
  if (a.epi != EP_STORE && a.epi != EP_RESID && a.epi != EP_GELU_MUL) return false;
  const int Kp = (a.K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const int MT = (a.M + 31) / 32;
  const size_t smem = ((size_t)Kp * MT * 32 + MT * 32) * sizeof(float);
  if (smem > 150 * 1024 || Kp / 2 < WS_WIN || a.k_split > 0 || (double)Kp * a.P * 4.0 >= 2147483648.0) return false;
  auto al8 = [](const TV& t) { return t.bs % 2 == 0 && (reinterpret_cast<uintptr_t>(t.p) & 7) == 0; };
  if (!al8(a.y) || ((a.epi == EP_RESID || a.epi == EP_GELU_MUL) && !al8(a.r))) return false;
  const int groups_per_img = (a.P + 63) / 64;
  const long total = (long)groups_per_img * a.B;
  // enough pixel groups that every wave gets one or two: the prefetch window then pays for itself
  // (3600 groups, a 640 x 360 plane: 86 against 75 TFLOP/s on the K-loop kernel)
  // enough pixel groups that every wave gets one or two: the prefetch window then pays for itself (3600 groups, a
  // 640 x 360 plane: 86 against 75 TFLOP/s on the K-loop kernel); from one group per SIMD on when the reduction is
  // long enough to amortise the weight load (1024 groups, a single 256 x 256 image: M96 K255 72 against 63 TFLOP/s,
  // M96 K96 35 against 41)
  // (between those, e.g. 2501 groups of a 488 x 328 plane, some SIMDs get two groups and most one: the K-loop
  // kernel's 96 x 128 tile wins there, 90 against 68 TFLOP/s at M96 K384)
  const long simd_rounds = (total + 1023) / 1024;
  const bool one_round = a.K >= 192 && total >= 1024 && (double)total >= 0.95 * 1024.0 * (double)simd_rounds && simd_rounds <= 2;
  if (total < 256L * WS_WAVES * 3 / 2 && !one_round) return false;
  const int grid = 256;
#define JIRF_WS_CASE(N, E)                                                                               \
  do {                                                                                                   \
    static DynLdsOnce once;                                                                              \
    allow_dynamic_lds(once, reinterpret_cast<const void*>(gemm_ws_kernel<N, E>), 152 * 1024);            \
    hipLaunchKernelGGL((gemm_ws_kernel<N, E>), dim3(grid), dim3(64 * WS_WAVES), smem, s, a, groups_per_img, (int)total); \
  } while (0)
#define JIRF_WS_EPI(N)                                                                                   \
  do {                                                                                                   \
    if (a.epi == EP_RESID) JIRF_WS_CASE(N, EP_RESID);                                                    \
    else if (a.epi == EP_GELU_MUL) JIRF_WS_CASE(N, EP_GELU_MUL);                                         \
    else JIRF_WS_CASE(N, EP_STORE);                                                                      \
  } while (0)
  if (MT == 1) JIRF_WS_EPI(1);
  else if (MT == 2) JIRF_WS_EPI(2);
  else JIRF_WS_EPI(3);
#undef JIRF_WS_EPI
#undef JIRF_WS_CASE
  JIRF_HIP_CHECK(hipGetLastError());
  return true;
}

// the split-bf16 K-loop kernel: long reductions on grids that fill the chip, shared weights
static bool t3_covers(const GemmArgs& a, int loader) {
  if (!gemm_split_bf16_enabled() || !a.wt3 || a.wt_bs != 0) return false;
  if (loader != LD_PLAIN && loader != LD_LN && loader != LD_CONV3) return false;
  if (loader == LD_LN && (!a.mu || !a.rstd || a.lnw)) return false;   // the affine part must be folded into the weights
  if (a.K <= 96) return false;
  if (loader == LD_CONV3) {
    if (a.M < 96 || a.Cin % 8 != 0 || (long)a.Cin * a.Hin * a.Win * 4 > 0x7fffffffL) return false;
  } else if (a.M < 128 || (long)a.K * a.P * 4 > 0x7fffffffL) {
    return false;
  }
  return (long)((a.M + T3_B - 1) / T3_B) * ((a.P + T3_B - 1) / T3_B) * a.B >= 512;
}
static void launch_t3(const GemmArgs& a, int loader, hipStream_t s) {
  const int tiles_m = (a.M + T3_B - 1) / T3_B, tiles_n = (a.P + T3_B - 1) / T3_B;
  const long blocks = (long)tiles_m * tiles_n * a.B;
  const dim3 grid((unsigned)blocks), block(256);
  if (loader == LD_LN) hipLaunchKernelGGL((gemm_t3_kernel<LD_LN>), grid, block, 0, s, a, tiles_m, tiles_n);
  else if (loader == LD_CONV3) hipLaunchKernelGGL((gemm_t3_kernel<LD_CONV3>), grid, block, 0, s, a, tiles_m, tiles_n);
  else hipLaunchKernelGGL((gemm_t3_kernel<LD_PLAIN>), grid, block, 0, s, a, tiles_m, tiles_n);
  JIRF_HIP_CHECK(hipGetLastError());
}

static bool xs_enabled() {
  static const bool on = !std::getenv("JIRF_GEMM_NO_XS") && !std::getenv("JIRF_GEMM_TILE");
  return on;
}
bool gemm_ln_inline(int K) { return K <= 256 && xs_enabled(); }
// (with the split-bf16 kernels on, a reduction past 96 on a grid that fills the chip goes to the K-loop form, which
//  reads the statistics: the same test as t3_covers, for a caller that has not built its GemmArgs yet)
bool gemm_ln_inline(int K, int M, int P, int B) {
  if (!gemm_ln_inline(K)) return false;
  const bool t3 = gemm_split_bf16_enabled() && K > 96 && M >= 128 && (long)K * P * 4 <= 0x7fffffffL &&
                  (long)((M + T3_B - 1) / T3_B) * ((P + T3_B - 1) / T3_B) * B >= 512;
  return !t3;
}

template <int BN, int TNW>
static void launch_xs(const GemmArgs& a_in, int loader, bool vec, hipStream_t s) {
#ifdef JIRF_XS_PROBE
  GemmArgs a_abl = a_in;
  if (loader == LD_LN && std::getenv("JIRF_XS_ABL")) a_abl.k_split = std::atoi(std::getenv("JIRF_XS_ABL"));
  const GemmArgs& a = a_abl;
#else
  const GemmArgs& a = a_in;
#endif
  const int tiles_n = (a.P + BN - 1) / BN;
  const int Kp = (a.K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const size_t smem = ((size_t)Kp * BN + 2 * BN + 2 * Kp + (a.bias ? (a.M + 31) / 32 * 32 : 0)) * sizeof(float);
  // the register-resident LayerNorm holds a lane's share of the tile in 16 slots
  JIRF_REQUIRE(loader != LD_LN || Kp <= 16 * (1024 / BN), JIRF_EINVAL, "launch_xs: K=%d does not fit %d-pixel tiles", a.K, BN);
  // fewer workgroups than twice the CUs: split the output-row tiles over gridDim.y (at least 4 per part)
  const int n_mt = (a.M + 31) / 32;
  int msplit = 1;
  while ((long)tiles_n * a.B * msplit < 512 && n_mt / (msplit * 2) >= 4) msplit *= 2;
  const dim3 grid((unsigned)(tiles_n * a.B), (unsigned)msplit), block(256);
#define JIRF_XS_CASE(LD, V)                                                                        \
  do {                                                                                             \
    static DynLdsOnce once;                                        \
    allow_dynamic_lds(once, reinterpret_cast<const void*>(gemm_xs_kernel<BN, TNW, LD, V>), 80 * 1024); \
    hipLaunchKernelGGL((gemm_xs_kernel<BN, TNW, LD, V>), grid, block, smem, s, a, tiles_n);        \
  } while (0)
  switch (loader) {
    case LD_PLAIN:
      if (vec) JIRF_XS_CASE(LD_PLAIN, true); else JIRF_XS_CASE(LD_PLAIN, false);
      break;
    case LD_GATHER:
      if (vec) JIRF_XS_CASE(LD_GATHER, true); else JIRF_XS_CASE(LD_GATHER, false);
      break;
    case LD_LN:
      if (vec) JIRF_XS_CASE(LD_LN, true); else JIRF_XS_CASE(LD_LN, false);
      break;
    default:
      JIRF_REQUIRE(false, JIRF_EINVAL, "launch_xs: loader %d", loader);
  }
#undef JIRF_XS_CASE
  JIRF_HIP_CHECK(hipGetLastError());
}

template <int BN>
static void launch_xs_bn(const GemmArgs& a, int loader, bool vec, hipStream_t s) {
  // pixel groups per output-row tile: keep the four waves evenly loaded
  const int n_mt = (a.M + 31) / 32;
  constexpr int TN = BN / 32;
  // (the attention * V epilogue is register-heavy: four accumulator tiles would spill)
  // (nine row tiles, M = 288: four-column items 3 / 2 / 2 / 2 over the waves beat two-column items 5 / 5 / 4 / 4,
  //  106 against 102 TFLOP/s - a weight fragment then feeds four MFMAs; at six row tiles the two-column form wins)
  if ((n_mt % 4 == 0 || n_mt >= 9) && loader != LD_GATHER)
    launch_xs<BN, TN>(a, loader, vec, s);
  else if ((n_mt * 2) % 4 == 0 || n_mt * 2 >= 12)
    launch_xs<BN, TN / 2>(a, loader, vec, s);
  else
    launch_xs<BN, 1>(a, loader, vec, s);
}

// Sums the split-K slabs in slab order (deterministic) and applies bias + epilogue.  One thread per output
// element; consecutive threads take consecutive pixels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a, const int splits) {
  const long n = (long)a.B * a.M * a.P;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int p = (int)(e % a.P);
  const long bm = e / a.P;
  const int m = (int)(bm % a.M), b = (int)(bm / a.M);
  float v = 0.f;
  for (int sp = 0; sp < splits; ++sp) v += a.splitk_ws[(long)sp * n + e];
  if (a.bias) v += a.bias[(long)b * a.bias_bs + m];
  float* __restrict__ yb = a.y.p + (long)b * a.y.bs;
  const long at = (long)m * a.P + p;
  switch (a.epi) {
    case EP_RESID: yb[at] = v + a.r.p[(long)b * a.r.bs + at]; break;
    case EP_GELU_MUL: yb[at] = gelu_erf(v) * a.r.p[(long)b * a.r.bs + at]; break;
    case EP_GELU: yb[at] = gelu_erf(v); break;
    case EP_LEAKY: yb[at] = v > 0.f ? v : 0.01f * v; break;
    case EP_UNSHUFFLE: {
      const int py = p / a.W, px = p - py * a.W;
      yb[(long)(4 * m + 2 * (py & 1) + (px & 1)) * (a.P >> 2) + (long)(py >> 1) * (a.W >> 1) + (px >> 1)] = v;
      break;
    }
    case EP_SHUFFLE: {
      const int py = p / a.W, px = p - py * a.W;
      yb[(long)(m >> 2) * (4L * a.P) + (long)(2 * py + ((m >> 1) & 1)) * (2 * a.W) + 2 * px + (m & 1)] = v;
      break;
    }
    default: yb[at] = v;
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
static void launch_tile(const GemmArgs& a, int loader, bool vec, hipStream_t s) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.P + BN - 1) / BN;
  // split-K: an underfilled grid with a long reduction is cut so that about two workgroups per CU exist
  // (at least 4 chunks of 16 per slab; the slabs must fit the caller's workspace)
  const long blocks = (long)tiles_m * tiles_n * a.B;
  const int nk = (a.K + GEMM_BK - 1) / GEMM_BK;
  int splits = 1;
  if (a.splitk_ws && loader != LD_GATHER && blocks < 256 && nk >= 8) {
    splits = (int)std::min<long>(std::min<long>(nk / 4, (512 + blocks - 1) / blocks), 32);
    while (splits > 1 && (size_t)splits * a.B * a.M * a.P > a.splitk_floats) --splits;
  }
  const int kt_per = (nk + splits - 1) / splits;
  splits = (nk + kt_per - 1) / kt_per;
  const dim3 grid((unsigned)blocks, (unsigned)splits), block(64 * WAVES_M * WAVES_N);
#define JIRF_GEMM_CASE(LD, V)                                                                   \
  hipLaunchKernelGGL((gemm_kernel<WAVES_M, WAVES_N, TM, TN, LD, V>), grid, block, 0, s, a, tiles_m, \
                     tiles_n, kt_per)
  switch (loader) {
    case LD_PLAIN:
      if (vec) JIRF_GEMM_CASE(LD_PLAIN, true); else JIRF_GEMM_CASE(LD_PLAIN, false);
      break;
    case LD_GATHER:
      if (vec) JIRF_GEMM_CASE(LD_GATHER, true); else JIRF_GEMM_CASE(LD_GATHER, false);
      break;
    case LD_LN:
      if (vec) JIRF_GEMM_CASE(LD_LN, true); else JIRF_GEMM_CASE(LD_LN, false);
      break;
    case LD_CONV3:
      JIRF_GEMM_CASE(LD_CONV3, false);
      break;
    default:
      JIRF_REQUIRE(false, JIRF_EINVAL, "launch_gemm: unknown loader %d", loader);
  }
#undef JIRF_GEMM_CASE
  JIRF_HIP_CHECK(hipGetLastError());
  if (splits > 1) {
    const long n = (long)a.B * a.M * a.P;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, splits);
    JIRF_HIP_CHECK(hipGetLastError());
  }
}

void launch_gemm(const GemmArgs& a_in, int loader, hipStream_t s) {
  GemmArgs a = a_in;
  if (loader == LD_CONV3) {
    if (a.ks == 0) { a.ks = 3; a.stride = 1; a.pad = 1; }
    if (a.Hin == 0) { a.Hin = a.H; a.Win = a.W; }
  }
  JIRF_REQUIRE(a.B > 0 && a.M > 0 && a.K > 0 && a.P > 0, JIRF_EINVAL,
               "launch_gemm: empty problem B=%d M=%d K=%d P=%d", a.B, a.M, a.K, a.P);
  JIRF_REQUIRE(a.Mpad % 4 == 0 && a.Mpad >= a.M, JIRF_EINVAL,
               "launch_gemm: packed row length %d does not cover M=%d", a.Mpad, a.M);
  if (loader == LD_CONV3)
    JIRF_REQUIRE(a.H * a.W == a.P && a.K == a.ks * a.ks * a.Cin && a.Hin < 16384 && a.Win < 32768 &&
                     a.ks >= 1 && a.ks <= 7 && a.stride >= 1 && a.pad >= 0 && a.pad <= 7 &&
                     (a.Hin + 2 * a.pad - a.ks) / a.stride + 1 == a.H && (a.Win + 2 * a.pad - a.ks) / a.stride + 1 == a.W,
                 JIRF_EINVAL, "launch_gemm: conv geometry out %dx%d in %dx%d k%d s%d p%d K=%d Cin=%d", a.H, a.W,
                 a.Hin, a.Win, a.ks, a.stride, a.pad, a.K, a.Cin);
  if (a.epi == EP_UNSHUFFLE || a.epi == EP_SHUFFLE)
    JIRF_REQUIRE(a.H * a.W == a.P && (a.epi == EP_SHUFFLE || (a.H % 2 == 0 && a.W % 2 == 0)),
                 JIRF_EINVAL, "launch_gemm: (un)shuffle geometry H=%d W=%d P=%d", a.H, a.W, a.P);
  const bool epi_ok =
      loader == LD_GATHER ? a.epi == EP_AV_INTRA
      : loader == LD_LN   ? a.epi == EP_STORE
      : loader == LD_CONV3 ? (a.epi == EP_STORE || a.epi == EP_RESID || a.epi == EP_UNSHUFFLE || a.epi == EP_SHUFFLE ||
                              a.epi == EP_LEAKY)
                           : (a.epi == EP_STORE || a.epi == EP_RESID || a.epi == EP_GELU_MUL || a.epi == EP_GELU ||
                              a.epi == EP_SHUFFLE);
  if (a.k_split > 0)
    JIRF_REQUIRE(loader == LD_PLAIN && a.x2.p && a.k_split <= a.K, JIRF_EINVAL, "launch_gemm: bad split source");
  JIRF_REQUIRE(epi_ok, JIRF_EINVAL, "launch_gemm: epilogue %d is not built for loader %d", a.epi, loader);
  // float4 activation loads need 16-byte aligned rows: pixel count and strides multiples of 4
  const bool vec = a.P % 4 == 0 && a.x.bs % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x.p) & 15) == 0 &&
                   (a.k_split == 0 || (a.x2.bs % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x2.p) & 15) == 0));
  const char* name = loader == LD_CONV3 ? "gemm_conv3x3" : loader == LD_GATHER ? "gemm_attn_v"
                     : loader == LD_LN  ? "gemm_ln_1x1" : "gemm_1x1";
  const double flops = 2.0 * a.M * (double)a.K * a.P * a.B;
  const double kin = loader == LD_CONV3 ? (double)a.Cin * a.Hin * a.Win / a.P : a.K;
  double bytes = 4.0 * a.B * (double)a.P * (kin + a.M) + 4.0 * a.K * (double)a.M * (a.wt_bs ? a.B : 1);
  if (a.epi == EP_RESID || a.epi == EP_GELU_MUL) bytes += 4.0 * a.B * (double)a.P * a.M;
  if (a.epi == EP_AV_INTRA) bytes += 4.0 * a.B * (double)a.P * a.M * 3;
  // which kernel serves the launch is part of the label: the split-bf16 forms are priced against another roof
  static const bool tile_forced_early = std::getenv("JIRF_GEMM_TILE") != nullptr;
  const bool xs_shape = xs_enabled() && loader != LD_CONV3 && a.K <= 256 &&
                        (loader == LD_LN || (a.K <= 128 && (a.M >= 2 * a.K || loader == LD_GATHER)));
  const int split_kind = tile_forced_early ? 0 : (xs_shape && gemm_b3_covers(a, loader)) ? 1 : t3_covers(a, loader) ? 2
                         : ((!xs_shape || loader == LD_GATHER) && gemm_k3_covers(a, loader)) ? 3 : 0;
  Scope sc(intern_label("%s%s M%d K%d P%d", name, split_kind ? "_bf16x3" : "", a.M, a.K, a.P), flops, bytes, s);

  // the RGB output convolutions: a handful of output rows would idle the matrix cores
  static const bool tile_forced = std::getenv("JIRF_GEMM_TILE") != nullptr;
  if (loader == LD_CONV3 && !tile_forced && try_launch_conv3x3_fewout(a, s)) return;

  // Kernel choice, from per-shape measurements on MI355X (HINT / ASTv2 / AST shapes, B = 16):
  //  * activation-stationary kernel when LayerNorm rides on the load (it also replaces the
  //    statistics pass), for attention * V with K <= 128, and for plain loads with a short K and
  //    M >= 2 K (each loaded tile then feeds a long enough MFMA phase);
  //  * otherwise the K-loop kernel, by default with its smallest 64 x 64 tile: 7 waves per SIMD hide
  //    the load / store phases of these bandwidth-heavy shapes better than bigger tiles do;
  //  * 128 x 128 only where the reduction is long and the output tall (MFMA-bound shapes),
  //    96 x 128 for M = 96 with K > 128.
  static const int forced = std::getenv("JIRF_GEMM_TILE") ? std::atoi(std::getenv("JIRF_GEMM_TILE")) : -1;
  const bool xs = xs_enabled() && loader != LD_CONV3 && a.K <= 256 &&
                  (loader == LD_LN || (a.K <= 128 && (a.M >= 2 * a.K || loader == LD_GATHER)));
  if (split_kind == 1) return launch_gemm_b3(a, loader, s);
  if (split_kind == 2) return launch_t3(a, loader, s);
  if (split_kind == 3) return launch_gemm_k3(a, loader, s);
  if (!xs && try_launch_ws(a, loader, vec, s)) return;
  if (xs) {
    // 128-pixel tiles need K <= 128 (LDS) and enough of them: an underfilled grid does better with 64-pixel
    // tiles (half the tile load per workgroup) than with more row splits of the wide ones
    const long wide_tiles = (long)((a.P + 127) / 128) * a.B;
    // the activation-stationary kernel's vector form also stores (and reads its epilogue operands) in
    // 16-byte pieces
    auto al16 = [](const TV& t) { return t.bs % 4 == 0 && (reinterpret_cast<uintptr_t>(t.p) & 15) == 0; };
    const bool vec_xs = vec && (a.epi == EP_SHUFFLE || al16(a.y)) &&
                        ((a.epi != EP_RESID && a.epi != EP_GELU_MUL) || al16(a.r)) &&
                        (a.epi != EP_AV_INTRA || (al16(a.y2) && al16(a.q) && al16(a.k)));
    if (a.K <= 128 && wide_tiles >= 512) launch_xs_bn<128>(a, loader, vec_xs, s);
    else launch_xs_bn<64>(a, loader, vec_xs, s);
    return;
  }
  if (loader == LD_LN)
    JIRF_REQUIRE(a.mu && a.rstd, JIRF_EINVAL, "launch_gemm: LayerNorm statistics are needed for K=%d", a.K);
  auto blocks = [&](int bm, int bn) {
    return (long)((a.M + bm - 1) / bm) * ((a.P + bn - 1) / bn) * a.B;
  };
  auto padded = [&](int bm) { return (a.M + bm - 1) / bm * bm; };
  if (forced == 0) return launch_tile<3, 1, 1, 4>(a, loader, vec, s);
  if (forced == 1) return launch_tile<2, 2, 2, 2>(a, loader, vec, s);
  if (forced == 2) return launch_tile<2, 2, 1, 2>(a, loader, vec, s);
  if (forced == 3) return launch_tile<2, 2, 1, 1>(a, loader, vec, s);
  // (per-shape sweep of the four tiles at B = 32, TFLOP/s as 96x128 / 128x128 / 64x128 / 64x64:
  //  1x1 M384 K1021 P1024 90 / 112 / 107 / 103, M384 K384 81 / 99 / 96 / 93, M192 K510 P4096 88 / 84 / 105 / 100,
  //  M192 K192 72 / 69 / 90 / 86, attention * V M192 K192 29 / 30 / 58 / 53 and M384 K384 40 / 45 / 71 / 69,
  //  3x3 M96 K1728 P4096 54 / 68 / 59 / 62,
  //  3x3 M192 K864 P16384 80 / 72 / 81 / 84)
  if (a.K >= 192 && a.M >= 384 && blocks(128, 128) >= 512 && (loader != LD_GATHER || a.M >= 768))
    launch_tile<2, 2, 2, 2>(a, loader, vec, s);
  else if (loader == LD_CONV3 && a.K >= 1024 && a.M >= 96 && blocks(128, 128) >= 512)
    launch_tile<2, 2, 2, 2>(a, loader, vec, s);
  else if (padded(96) < padded(64) && (a.K > 128 || (a.K >= 96 && blocks(96, 128) >= 1024)) && blocks(96, 128) >= 512)
    launch_tile<3, 1, 1, 4>(a, loader, vec, s);
  else if (loader != LD_CONV3 && a.K >= 192 && a.M >= 128 && blocks(64, 128) >= 1024)
    launch_tile<2, 2, 1, 2>(a, loader, vec, s);
  // long reductions on small grids (the deep levels of a single image or of a training batch of two; split-K supplies
  // the workgroups): wider tiles re-read the operands less - 1x1 M512 K4096 P1024 at B = 2 50 -> 84 TFLOP/s with
  // 64 x 128, 3x3 M512 K2304 P1024 34 -> 54 with 128 x 128
  else if (loader != LD_CONV3 && a.K >= 1024 && a.M >= 128 && a.P >= 1024)
    launch_tile<2, 2, 1, 2>(a, loader, vec, s);
  else if (loader == LD_CONV3 && a.K >= 2048 && a.M >= 256)
    launch_tile<2, 2, 2, 2>(a, loader, vec, s);
  else
    launch_tile<2, 2, 1, 1>(a, loader, vec, s);
}

int pack_wt_1x1(const float* w, int M, int K, std::vector<float>& out, int* Mpad) {
  const int mp = (M + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
  const int kp = (K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  out.assign((size_t)kp * mp, 0.f);
  for (int o = 0; o < M; ++o)
    for (int k = 0; k < K; ++k) out[(size_t)k * mp + o] = w[(size_t)o * K + k];
  *Mpad = mp;
  return kp;
}

// w [M][Cin][ks][ks] -> Wt[(tap * Cin + c)][o], tap = ky * ks + kx
int pack_wt_conv(const float* w, int M, int Cin, int ks, std::vector<float>& out, int* Mpad) {
  const int taps = ks * ks, K = taps * Cin;
  const int mp = (M + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
  const int kp = (K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  out.assign((size_t)kp * mp, 0.f);
  for (int o = 0; o < M; ++o)
    for (int c = 0; c < Cin; ++c)
      for (int t = 0; t < taps; ++t)
        out[(size_t)(t * Cin + c) * mp + o] = w[((size_t)o * Cin + c) * taps + t];
  *Mpad = mp;
  return kp;
}

int pack_wt_3x3(const float* w, int M, int Cin, std::vector<float>& out, int* Mpad) {
  return pack_wt_conv(w, M, Cin, 3, out, Mpad);
}

}  // namespace jirf
