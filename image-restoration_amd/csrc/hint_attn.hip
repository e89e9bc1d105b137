// HINT channel attention (HMHA) statistics and C x C micro-kernels for gfx950.
//
// Reference semantics: Attention.execute / ReGroup / Inter_CacheModulation /
// calculate_layer_cache / AdaptiveAvgPool2d in basicsr/models/archs/hint_arch.py:59-95,189-210,
// 231-296,345-399.
//
// Pipeline per attention block (q, k are channel slices [C][P] of the depthwise-conv output):
//   1. chan_stats   one pass over q and k: split-K contraction over the pixel axis producing
//                   partial Q.K^T (fp32 MFMA), partial Q.Q^T and row sums (fp64 MFMA — these feed
//                   an argsort, see below), partial sum q^2 / sum k^2.
//   2. regroup_rows deterministic reduction of the partial slabs; row i of the channel
//                   correlation matrix -> mean similarity; reciprocal L2 norms.
//   3. group_attn   one workgroup per (channel group, image): ranks the similarities (the channel
//                   permutation), gathers its c x c block of Q.K^T, scales by the norms and the
//                   temperature, row softmax, inter-cache modulation, writes the block both in the
//                   GEMM weight layout (for attention * V) and in natural orientation.
//   4. cache_update bilinear upsample of every group's (batch-mean) attention to C x C, float
//                   floor-division, momentum update of qv_cache.
//
// Why fp64 for Q.Q^T: the channel order is an argsort of the correlation row means, and adjacent
// means are routinely 1e-7 apart at C = 384.  fp64 accumulation makes the computed means the
// rounding-free value of the reference's formula, so ties are decided by the data, not by the
// summation order.
#include "common.h"

namespace jirf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int ST_T = 64;    // channel tile edge of the statistics kernel
constexpr int ST_CH = 32;   // pixels per staged chunk
constexpr int ST_LD = 33;   // LDS row stride (floats)
constexpr float NORM_EPS = 1e-30f;

__device__ __forceinline__ float gelu_erf_a(float v) { return gelu_fast(v); }

// ------------------------------------------------------------------------------------------------
// 1. statistics
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
  const float* q; long q_bs;     // per-image q (Q.K^T rows, sum q^2)
  const float* k; long k_bs;     // per-image k
  const float* qg; long qg_bs;   // gram source: q of the image (per-image mode) or batch-mean q
  int C, P, tiles, splits, chunks_per_split;
  int head_ch;       // > 0: only tile pairs that intersect a diagonal head block of this size
  float* qk_part;    // [nb][splits][C][C]
  double* gram_part; // [nb][splits][C][C]
  double* qsum_part; // [nb][splits][C]
  float* q2_part;    // [nb][splits][C]
  float* k2_part;    // [nb][splits][C]
};

// Tiles staged per 32-pixel chunk (64 channel rows each, row stride 33 floats):
//   A  rows of the i tile of q (of the gram source when only the gram is computed)
//   B  rows of the j tile of k                       (DO_QK)
//   Cg rows of the j tile of the gram source         (DO_GRAM; aliases A on diagonal tile pairs)
// The next chunk is fetched into registers while the current one is multiplied (LDS double
// buffer, one barrier per chunk).
template <bool DO_QK, bool DO_GRAM, bool VEC>
__global__ __launch_bounds__(256) void chan_stats_kernel(const StatsArgs a) {
  __shared__ float TA[2][ST_T * ST_LD];
  __shared__ float TB[DO_QK ? 2 : 1][DO_QK ? ST_T * ST_LD : 1];
  __shared__ float TC[DO_GRAM ? 2 : 1][DO_GRAM ? ST_T * ST_LD : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
  const int ti = blockIdx.x / a.tiles, tj = blockIdx.x % a.tiles;
  const int sp = blockIdx.y, b = blockIdx.z;
  const int C = a.C, P = a.P;
  const int i0 = ti * ST_T, j0 = tj * ST_T;
  if (a.head_ch > 0) {
    // heads touched by the row tile / by the column tile; nothing to do when they are disjoint
    const int hi0 = i0 / a.head_ch, hi1 = (min(i0 + ST_T, C) - 1) / a.head_ch;
    const int hj0 = j0 / a.head_ch, hj1 = (min(j0 + ST_T, C) - 1) / a.head_ch;
    if (hi1 < hj0 || hj1 < hi0) return;
  }
  const int c_begin = sp * a.chunks_per_split;
  const int c_end = min(c_begin + a.chunks_per_split, (P + ST_CH - 1) / ST_CH);
  // an empty split (c_begin >= c_end) still writes its all-zero slab: the reduction reads every slab

  const float* __restrict__ src_a = (DO_QK ? a.q + (long)b * a.q_bs : a.qg + (long)b * a.qg_bs);
  const float* __restrict__ src_b = a.k + (long)b * a.k_bs;
  const float* __restrict__ src_c = a.qg + (long)b * a.qg_bs;
  const bool c_is_a = DO_GRAM && ti == tj && src_c == src_a;

  f32x16 s_acc;
  f64x4 g_acc[2][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) g_acc[t][u][r] = 0.0;
  double row_sum = 0.0;  // threads 128..191: sum of the gram-source row i0 + tid - 128
  float row_sq = 0.f;    // threads 0..63: sum q^2 of row i0 + tid; threads 64..127: sum k^2 of row j0 + tid - 64

  // 64 rows x 32 px = 512 float4 slots per tile, two per thread
  auto fetch = [&](const float* __restrict__ src, int r0, int px0, float4* out) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int s = tid + it * 256;
      const int row = r0 + (s >> 3), px = px0 + (s & 7) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < C) {
        const float* __restrict__ ptr = src + (long)row * P + px;
        if constexpr (VEC) {
          if (px < P) v = *reinterpret_cast<const float4*>(ptr);
        } else {
          if (px < P) v.x = ptr[0];
          if (px + 1 < P) v.y = ptr[1];
          if (px + 2 < P) v.z = ptr[2];
          if (px + 3 < P) v.w = ptr[3];
        }
      }
      out[it] = v;
    }
  };
  auto put = [&](float* dst, const float4* in) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int s = tid + it * 256;
      float* __restrict__ d = dst + (s >> 3) * ST_LD + (s & 7) * 4;
      d[0] = in[it].x; d[1] = in[it].y; d[2] = in[it].z; d[3] = in[it].w;
    }
  };
  float4 ra[2], rb[2], rc[2];
  auto fetch_all = [&](int ch) {
    const int px0 = ch * ST_CH;
    fetch(src_a, i0, px0, ra);
    if constexpr (DO_QK) fetch(src_b, j0, px0, rb);
    if constexpr (DO_GRAM) {
      if (!c_is_a) fetch(src_c, j0, px0, rc);
    }
  };
  auto put_all = [&](int buf) {
    put(TA[buf], ra);
    if constexpr (DO_QK) put(TB[buf], rb);
    if constexpr (DO_GRAM) {
      if (!c_is_a) put(TC[buf], rc);
    }
  };

  if (c_begin < c_end) {
    fetch_all(c_begin);
    put_all(0);
  }
  __syncthreads();
  const bool quad_on = i0 + wi * 32 < C && j0 + wj * 32 < C;
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) fetch_all(ch + 1);
    const float* __restrict__ A = TA[buf];
    if constexpr (DO_QK) {
      const float* __restrict__ Bt = TB[buf];
      if (quad_on) {
        const float* __restrict__ ap = &A[(wi * 32 + l31) * ST_LD + half];
        const float* __restrict__ bp = &Bt[(wj * 32 + l31) * ST_LD + half];
#pragma unroll
        for (int s = 0; s < ST_CH / 2; ++s)
          s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], s_acc, 0, 0, 0);
      }
      if (tj == ti && tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < ST_CH; ++e) t = fmaf(A[tid * ST_LD + e], A[tid * ST_LD + e], t);
        row_sq += t;
      }
      if (ti == tj && tid >= 64 && tid < 128) {
        const int r = tid - 64;
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < ST_CH; ++e) t = fmaf(Bt[r * ST_LD + e], Bt[r * ST_LD + e], t);
        row_sq += t;
      }
    }
    if constexpr (DO_GRAM) {
      const float* __restrict__ Ct = c_is_a ? A : TC[buf];
      if (quad_on) {
#pragma unroll
        for (int s = 0; s < ST_CH / 4; ++s) {
          double av[2], bv[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            av[t] = (double)A[(wi * 32 + t * 16 + l15) * ST_LD + 4 * s + quad];
            bv[t] = (double)Ct[(wj * 32 + t * 16 + l15) * ST_LD + 4 * s + quad];
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              g_acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[u], g_acc[t][u], 0, 0, 0);
        }
      }
      if (tj == ti && tid >= 128 && tid < 192) {
        const int r = tid - 128;
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < ST_CH; ++e) t += (double)A[r * ST_LD + e];
        row_sum += t;
      }
    }
    if (ch + 1 < c_end) put_all(buf ^ 1);
    __syncthreads();
  }

  const long slab = ((long)b * a.splits + sp) * C;
  if constexpr (DO_QK) {
    float* __restrict__ out = a.qk_part + slab * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int col = j0 + wj * 32 + l31;
      if (row < C && col < C) out[(long)row * C + col] = s_acc[r];
    }
    if (tj == ti && tid < 64 && i0 + tid < C) a.q2_part[slab + i0 + tid] = row_sq;
    if (ti == tj && tid >= 64 && tid < 128 && j0 + tid - 64 < C) a.k2_part[slab + j0 + tid - 64] = row_sq;
  }
  if constexpr (DO_GRAM) {
    double* __restrict__ out = a.gram_part + slab * C;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + wi * 32 + t * 16 + quad + 4 * r;
          const int col = j0 + wj * 32 + u * 16 + l15;
          if (row < C && col < C) out[(long)row * C + col] = g_acc[t][u][r];
        }
    if (tj == ti && tid >= 128 && tid < 192 && i0 + tid - 128 < C)
      a.qsum_part[slab + i0 + tid - 128] = row_sum;
  }
}

template <bool DO_QK, bool DO_GRAM>
static void launch_chan_stats(const StatsArgs& st, int nb, hipStream_t s) {
  auto ok = [&](const float* p, long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && bs % 4 == 0; };
  bool vec = st.P % 4 == 0;
  if (DO_QK) vec = vec && ok(st.q, st.q_bs) && ok(st.k, st.k_bs);
  if (DO_GRAM) vec = vec && ok(st.qg, st.qg_bs);
  const dim3 grid(st.tiles * st.tiles, st.splits, nb), block(256);
  if (vec)
    hipLaunchKernelGGL((chan_stats_kernel<DO_QK, DO_GRAM, true>), grid, block, 0, s, st);
  else
    hipLaunchKernelGGL((chan_stats_kernel<DO_QK, DO_GRAM, false>), grid, block, 0, s, st);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- 96-channel super-tile form (HINT) ------------------------------------------------------------
// HINT's channel counts are 48 * 2^k, so 64 x 64 tiles leave a third of the matrix cores idle at
// C = 96 and load the four waves unevenly.  Here a workgroup owns a 96 x 96 tile pair, split into
// 3 x 3 quadrants of 32 x 32; the quadrant products (Q.K^T: fp32 MFMA, cost 1; Q.Q^T: fp64 MFMA,
// cost 2) are dealt to the four waves greedily by cost, each wave keeping up to three of each kind
// in registers.  Q.Q^T is symmetric: only quadrants on or above the diagonal are computed (tile
// pairs with ti > tj compute none) and written to both (i, j) and (j, i).
constexpr int SQ_T = 96;
constexpr int SQ_SLOTS = 3;

template <bool DO_QK, bool DO_GRAM, bool VEC>
__global__ __launch_bounds__(256, 2) void chan_stats96_kernel(const StatsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sq_smem[];
  constexpr int TILE = SQ_T * ST_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
  const int ti = blockIdx.x / a.tiles, tj = blockIdx.x % a.tiles;
  const int sp = blockIdx.y, b = blockIdx.z;
  const int C = a.C, P = a.P;
  const int i0 = ti * SQ_T, j0 = tj * SQ_T;
  const bool diag = ti == tj;
  const bool gram_on = DO_GRAM && ti <= tj;
  if (!DO_QK && !gram_on) return;
  // fixed heads of head_ch channels (head_ch divides the tile edge): only the diagonal tile pairs
  // hold head blocks, and of their quadrants only those that meet one
  const int hch = DO_GRAM ? 0 : a.head_ch;
  if (hch > 0 && !diag) return;
  const int c_begin = sp * a.chunks_per_split;
  const int c_end = min(c_begin + a.chunks_per_split, (P + ST_CH - 1) / ST_CH);

  const float* __restrict__ src_a = (DO_QK ? a.q + (long)b * a.q_bs : a.qg + (long)b * a.qg_bs);
  const float* __restrict__ src_b = a.k + (long)b * a.k_bs;
  const float* __restrict__ src_c = a.qg + (long)b * a.qg_bs;
  const bool c_is_a = gram_on && diag && src_c == src_a;
  const bool need_c = gram_on && !c_is_a;
  float* TA = sq_smem;                          // [2][TILE]
  float* TB = TA + 2 * TILE;                    // [2][TILE]  (DO_QK)
  float* TC = TB + (DO_QK ? 2 * TILE : 0);      // [2][TILE]  (gram source rows of tile tj)

  // ---- deal the quadrants --------------------------------------------------------------------
  const int nr = min(3, (C - i0 + 31) >> 5), nc = min(3, (C - j0 + 31) >> 5);
  int load[4] = {0, 0, 0, 0};
  int gcode[SQ_SLOTS], scode[SQ_SLOTS];
#pragma unroll
  for (int i = 0; i < SQ_SLOTS; ++i) gcode[i] = scode[i] = -1;
  auto least = [&]() {
    int w = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (load[i] < load[w]) w = i;
    return w;
  };
  auto give = [&](int w, int cost) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i == w) load[i] += cost;
  };
  auto push = [&](int* slots, int code) {
    if (slots[0] < 0) slots[0] = code;
    else if (slots[1] < 0) slots[1] = code;
    else slots[2] = code;
  };
  if (gram_on) {
#pragma unroll
    for (int it = 0; it < 9; ++it) {
      const int qi = it / 3, qj = it % 3;
      if (qi < nr && qj < nc && (!diag || qi <= qj)) {
        const int w = least();
        if (w == wave) push(gcode, qi * 4 + qj);
        give(w, 2);
      }
    }
  }
  if constexpr (DO_QK) {
#pragma unroll
    for (int it = 0; it < 9; ++it) {
      const int qi = it / 3, qj = it % 3;
      bool needed = qi < nr && qj < nc;
      if (hch > 0 && needed) {   // heads met by the quadrant's rows and by its columns must overlap
        const int r_lo = (qi * 32) / hch, r_hi = min(qi * 32 + 31, SQ_T - 1) / hch;
        const int c_lo = (qj * 32) / hch, c_hi = min(qj * 32 + 31, SQ_T - 1) / hch;
        needed = max(r_lo, c_lo) <= min(r_hi, c_hi);
      }
      if (needed) {
        const int w = least();
        if (w == wave) push(scode, qi * 4 + qj);
        give(w, 1);
      }
    }
  }

  f32x16 s_acc[SQ_SLOTS];
  f64x4 g_acc[SQ_SLOTS][2][2];
#pragma unroll
  for (int i = 0; i < SQ_SLOTS; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[i][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) g_acc[i][t][u][r] = 0.0;
  }
  double row_sum = 0.0;  // threads 0..95: sum of the gram-source row i0 + tid   (diagonal pairs)
  float row_sq = 0.f;    // threads 0..95: sum q^2 of row i0 + tid; 96..191: sum k^2 of row j0 + tid - 96

  // 96 rows x 32 px = 768 float4 slots per tile, three per thread
  auto fetch = [&](const float* __restrict__ src, int r0, int px0, float4* out) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int sl = tid + it * 256;
      const int row = r0 + (sl >> 3), px = px0 + (sl & 7) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < C) {
        const float* __restrict__ ptr = src + (long)row * P + px;
        if constexpr (VEC) {
          if (px < P) v = *reinterpret_cast<const float4*>(ptr);
        } else {
          if (px < P) v.x = ptr[0];
          if (px + 1 < P) v.y = ptr[1];
          if (px + 2 < P) v.z = ptr[2];
          if (px + 3 < P) v.w = ptr[3];
        }
      }
      out[it] = v;
    }
  };
  auto put = [&](float* dst, const float4* in) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int sl = tid + it * 256;
      float* __restrict__ d = dst + (sl >> 3) * ST_LD + (sl & 7) * 4;
      d[0] = in[it].x; d[1] = in[it].y; d[2] = in[it].z; d[3] = in[it].w;
    }
  };
  float4 ra[3], rb[3], rc[3];
  auto fetch_all = [&](int ch) {
    const int px0 = ch * ST_CH;
    fetch(src_a, i0, px0, ra);
    if constexpr (DO_QK) fetch(src_b, j0, px0, rb);
    if (need_c) fetch(src_c, j0, px0, rc);
  };
  auto put_all = [&](int buf) {
    put(TA + buf * TILE, ra);
    if constexpr (DO_QK) put(TB + buf * TILE, rb);
    if (need_c) put(TC + buf * TILE, rc);
  };

  if (c_begin < c_end) {
    fetch_all(c_begin);
    put_all(0);
  }
  __syncthreads();
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int buf = (ch - c_begin) & 1;
    if (ch + 1 < c_end) fetch_all(ch + 1);
    const float* __restrict__ A = TA + buf * TILE;
    const float* __restrict__ Bt = TB + buf * TILE;
    const float* __restrict__ Ct = c_is_a ? A : TC + buf * TILE;
    if constexpr (DO_QK) {
#pragma unroll
      for (int i = 0; i < SQ_SLOTS; ++i) {
        if (scode[i] < 0) continue;
        const float* __restrict__ ap = &A[((scode[i] >> 2) * 32 + l31) * ST_LD + half];
        const float* __restrict__ bp = &Bt[((scode[i] & 3) * 32 + l31) * ST_LD + half];
#pragma unroll
        for (int k2 = 0; k2 < ST_CH / 2; ++k2)
          s_acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * k2], bp[2 * k2], s_acc[i], 0, 0, 0);
      }
      if (diag && tid < 2 * SQ_T) {
        const float* __restrict__ rowp = tid < SQ_T ? &A[tid * ST_LD] : &Bt[(tid - SQ_T) * ST_LD];
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < ST_CH; ++e) t = fmaf(rowp[e], rowp[e], t);
        row_sq += t;
      }
    }
    if (gram_on) {
#pragma unroll
      for (int i = 0; i < SQ_SLOTS; ++i) {
        if (gcode[i] < 0) continue;
        const float* __restrict__ ap = &A[((gcode[i] >> 2) * 32 + l15) * ST_LD + quad];
        const float* __restrict__ cp = &Ct[((gcode[i] & 3) * 32 + l15) * ST_LD + quad];
#pragma unroll
        for (int k4 = 0; k4 < ST_CH / 4; ++k4) {
          double av[2], bv[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            av[t] = (double)ap[t * 16 * ST_LD + 4 * k4];
            bv[t] = (double)cp[t * 16 * ST_LD + 4 * k4];
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              g_acc[i][t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[u], g_acc[i][t][u], 0, 0, 0);
        }
      }
      if (diag && tid < SQ_T) {
        const float* __restrict__ rowp = &Ct[tid * ST_LD];
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < ST_CH; ++e) t += (double)rowp[e];
        row_sum += t;
      }
    }
    if (ch + 1 < c_end) put_all(buf ^ 1);
    __syncthreads();
  }

  const long slab = ((long)b * a.splits + sp) * C;
  if constexpr (DO_QK) {
    float* __restrict__ out = a.qk_part + slab * C;
#pragma unroll
    for (int i = 0; i < SQ_SLOTS; ++i) {
      if (scode[i] < 0) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + (scode[i] >> 2) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int col = j0 + (scode[i] & 3) * 32 + l31;
        if (row < C && col < C) out[(long)row * C + col] = s_acc[i][r];
      }
    }
    if (diag && tid < SQ_T && i0 + tid < C) a.q2_part[slab + i0 + tid] = row_sq;
    if (diag && tid >= SQ_T && tid < 2 * SQ_T && j0 + tid - SQ_T < C) a.k2_part[slab + j0 + tid - SQ_T] = row_sq;
  }
  if (gram_on) {
    double* __restrict__ out = a.gram_part + slab * C;
#pragma unroll
    for (int i = 0; i < SQ_SLOTS; ++i) {
      if (gcode[i] < 0) continue;
      // quadrants off the diagonal also fill their mirror image, so the reduction reads rows
      const bool mirror = !diag || (gcode[i] >> 2) != (gcode[i] & 3);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i0 + (gcode[i] >> 2) * 32 + t * 16 + quad + 4 * r;
            const int col = j0 + (gcode[i] & 3) * 32 + u * 16 + l15;
            if (row < C && col < C) {
              out[(long)row * C + col] = g_acc[i][t][u][r];
              if (mirror) out[(long)col * C + row] = g_acc[i][t][u][r];
            }
          }
    }
    if (diag && tid < SQ_T && i0 + tid < C) a.qsum_part[slab + i0 + tid] = row_sum;
  }
}

template <bool DO_QK, bool DO_GRAM>
static void launch_chan_stats96(const StatsArgs& st, int nb, hipStream_t s) {
  auto ok = [&](const float* p, long bs) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && bs % 4 == 0; };
  bool vec = st.P % 4 == 0;
  if (DO_QK) vec = vec && ok(st.q, st.q_bs) && ok(st.k, st.k_bs);
  if (DO_GRAM) vec = vec && ok(st.qg, st.qg_bs);
  const dim3 grid(st.tiles * st.tiles, st.splits, nb), block(256);
  // A (+ B) (+ C when a tile pair is off the diagonal or the gram source differs from q)
  const bool need_c = DO_GRAM && (st.tiles > 1 || (DO_QK && st.qg != st.q));
  const size_t smem = (size_t)(2 + (DO_QK ? 2 : 0) + (need_c ? 2 : 0)) * SQ_T * ST_LD * sizeof(float);
#define JIRF_SQ_CASE(V)                                                                              \
  do {                                                                                               \
    static DynLdsOnce once;                                        \
    allow_dynamic_lds(once, reinterpret_cast<const void*>(chan_stats96_kernel<DO_QK, DO_GRAM, V>), 80 * 1024); \
    hipLaunchKernelGGL((chan_stats96_kernel<DO_QK, DO_GRAM, V>), grid, block, smem, s, st);          \
  } while (0)
  if (vec) JIRF_SQ_CASE(true); else JIRF_SQ_CASE(false);
#undef JIRF_SQ_CASE
  JIRF_HIP_CHECK(hipGetLastError());
}

// batch mean of q for the coupled mode: qmean[c][p] = mean_b q[b][c][p]
__global__ __launch_bounds__(256) void batch_mean_kernel(const float* __restrict__ q, long q_bs,
                                                         float* __restrict__ out, int B, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += q[(long)b * q_bs + i];
  out[i] = s / (float)B;
}

// ------------------------------------------------------------------------------------------------
// 2. slab reduction, correlation row means, reciprocal norms
// ------------------------------------------------------------------------------------------------
// Sums the split-K slabs of every matrix / vector element into slab 0.  A workgroup takes 256 / PARTS
// elements with PARTS threads per element: thread `part` adds slabs part, part + PARTS, ... in order, the
// partial sums are then added in order, so the result does not depend on scheduling.
template <typename T, int PARTS>
__device__ __forceinline__ void reduce_slabs_part(T* __restrict__ base, long stride, int splits,
                                                  int part, T* __restrict__ red, int el, bool on) {
  constexpr int EL = 256 / PARTS;
  T t = 0;
  if (on) {
    // four loads in flight per thread (clamped, unpredicated), added in slab order
    for (int sp0 = part; sp0 < splits; sp0 += 4 * PARTS) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = base[(long)min(sp0 + PARTS * u, splits - 1) * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (sp0 + PARTS * u < splits) t += v[u];
    }
  }
  red[part * EL + el] = t;
  __syncthreads();
  if (on && part == 0) {
    T u = 0;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) u += red[q * EL + el];
    base[0] = u;
  }
  __syncthreads();
}

template <int PARTS>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* __restrict__ qk_part,
                                                           double* __restrict__ gram_part,
                                                           double* __restrict__ qsum_part,
                                                           float* __restrict__ q2_part,
                                                           float* __restrict__ k2_part, int C,
                                                           int splits, int B, int G) {
  constexpr int EL = 256 / PARTS;
  __shared__ double red_d[256];
  __shared__ float red_f[256];
  const int el = threadIdx.x % EL, part = threadIdx.x / EL;
  const long e0 = (long)blockIdx.x * EL, e = e0 + el;
  const int b = blockIdx.y;
  const long CC = (long)C * C;
  // matrices: elements [0, CC); vectors: elements [CC, CC + C).  A workgroup may straddle the boundary, so
  // each kind has a per-thread mask; kinds with no element in this workgroup are skipped (uniform branch).
  const bool mat = e < CC, vec = e >= CC && e < CC + C;
  const bool any_mat = e0 < CC, any_vec = e0 + EL > CC;
  const long c = e - CC;
  if (any_mat) {
    reduce_slabs_part<float, PARTS>(qk_part + (long)b * splits * CC + (mat ? e : 0), CC, splits, part, red_f, el, mat && b < B);
    if (gram_part)
      reduce_slabs_part<double, PARTS>(gram_part + (long)(b < G ? b : 0) * splits * CC + (mat ? e : 0), CC, splits, part, red_d, el,
                                       mat && b < G);
  }
  if (any_vec) {
    reduce_slabs_part<float, PARTS>(q2_part + (long)b * splits * C + (vec ? c : 0), C, splits, part, red_f, el, vec && b < B);
    reduce_slabs_part<float, PARTS>(k2_part + (long)b * splits * C + (vec ? c : 0), C, splits, part, red_f, el, vec && b < B);
    if (qsum_part)
      reduce_slabs_part<double, PARTS>(qsum_part + (long)(b < G ? b : 0) * splits * C + (vec ? c : 0), C, splits, part, red_d, el,
                                       vec && b < G);
  }
}

static void launch_reduce_slabs(float* qk_part, double* gram_part, double* qsum_part, float* q2_part,
                                float* k2_part, int C, int splits, int B, int G, hipStream_t s) {
  if (splits <= 1) return;
  const long n = (long)C * C + C;
  // (32 threads per element, 8 elements per workgroup, was measured slower at 256 slabs: 32-byte segments)
  if (splits <= 16)   // large batches: few slabs, many images - two threads per element, 128 elements per workgroup
    hipLaunchKernelGGL(reduce_slabs_kernel<2>, dim3((unsigned)((n + 127) / 128), std::max(B, G)), dim3(256), 0, s,
                       qk_part, gram_part, qsum_part, q2_part, k2_part, C, splits, B, G);
  else
    hipLaunchKernelGGL(reduce_slabs_kernel<8>, dim3((unsigned)((n + 31) / 32), std::max(B, G)), dim3(256), 0, s,
                       qk_part, gram_part, qsum_part, q2_part, k2_part, C, splits, B, G);
  JIRF_HIP_CHECK(hipGetLastError());
}

// One wave per channel row i (four rows per workgroup): sums the slabs of row i of Q K^T, the reciprocal
// norms, and (when sim != null) the mean of row i of the correlation matrix in double.  The per-channel
// means and standard deviations are formed once per workgroup by all of its threads.
constexpr int RG_ROWS = 4;
__global__ __launch_bounds__(256) void regroup_rows_kernel(
    const float* __restrict__ qk_part, const double* __restrict__ gram_part,
    const double* __restrict__ qsum_part, const float* __restrict__ q2_part,
    const float* __restrict__ k2_part, float* __restrict__ s_qk, float* __restrict__ rq,
    float* __restrict__ rk, double* __restrict__ sim, int C, int P, int splits, int nsum, int ipg,
    float norm_add) {
  extern __shared__ double dsm[];  // mean[C], dstd[C]
  double* mean = dsm;
  double* dstd = dsm + C;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * RG_ROWS + wave;
  const bool row_on = i < C;

  if (row_on) {
    for (int j = lane; j < C; j += 64) {
      float s = 0.f;
#pragma unroll 8
      for (int sp = 0; sp < nsum; ++sp) s += qk_part[(((long)b * splits + sp) * C + i) * C + j];
      s_qk[((long)b * C + i) * C + j] = s;
    }
    if (lane == 0) {
      float q2 = 0.f, k2 = 0.f;
#pragma unroll 8
      for (int sp = 0; sp < nsum; ++sp) {
        q2 += q2_part[((long)b * splits + sp) * C + i];
        k2 += k2_part[((long)b * splits + sp) * C + i];
      }
      if (norm_add > 0.f) {
        rq[(long)b * C + i] = 1.0f / (sqrtf(q2) + norm_add);
        rk[(long)b * C + i] = 1.0f / (sqrtf(k2) + norm_add);
      } else {
        rq[(long)b * C + i] = 1.0f / sqrtf(fmaxf(q2, NORM_EPS));
        rk[(long)b * C + i] = 1.0f / sqrtf(fmaxf(k2, NORM_EPS));
      }
    }
  }
  if (sim == nullptr || b % ipg != 0) return;   // uniform over the workgroup
  const int g = b / ipg;
  const double L = (double)P;
  for (int j = tid; j < C; j += 256) {
    double qs = 0.0, gjj = 0.0;
#pragma unroll 8
    for (int sp = 0; sp < nsum; ++sp) {
      qs += qsum_part[((long)g * splits + sp) * C + j];
      gjj += gram_part[(((long)g * splits + sp) * C + j) * C + j];
    }
    const double m = qs / L;
    mean[j] = m;
    dstd[j] = sqrt((gjj - L * m * m) / (L - 1.0));
  }
  __syncthreads();
  if (!row_on) return;
  double acc = 0.0;
  const double mi = mean[i], di = dstd[i];
  for (int j = lane; j < C; j += 64) {
    double gij = 0.0;
#pragma unroll 8
    for (int sp = 0; sp < nsum; ++sp) gij += gram_part[(((long)g * splits + sp) * C + i) * C + j];
    acc += ((gij - L * mi * mean[j]) / (L - 1.0)) / (di * dstd[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) sim[(long)g * C + i] = acc / (double)C;
}

// ------------------------------------------------------------------------------------------------
// 0. cache pooling  (AdaptiveAvgPool2d of the carried cache to this block's C)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pool_start(int i, int n_in, int n_out) { return (i * n_in) / n_out; }
__device__ __forceinline__ int pool_end(int i, int n_in, int n_out) {
  return ((i + 1) * n_in + n_out - 1) / n_out;
}

__global__ __launch_bounds__(256) void cache_pool_kernel(const float* __restrict__ in, int Cin,
                                                         float* __restrict__ out, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (idx >= C * C) return;
  const int i = idx / C, j = idx - i * C;
  const float* __restrict__ src = in + (long)g * Cin * Cin;
  if (Cin == C) {
    out[(long)g * C * C + idx] = src[idx];
    return;
  }
  const int ys = pool_start(i, Cin, C), ye = pool_end(i, Cin, C);
  const int xs = pool_start(j, Cin, C), xe = pool_end(j, Cin, C);
  double s = 0.0;
  for (int y = ys; y < ye; ++y)
    for (int x = xs; x < xe; ++x) s += (double)src[(long)y * Cin + x];
  out[(long)g * C * C + idx] = (float)(s / (double)((ye - ys) * (xe - xs)));
}

// ------------------------------------------------------------------------------------------------
// 3. per-group attention
// ------------------------------------------------------------------------------------------------
struct GroupAttnArgs {
  int C, G, ipg, Mpad;
  int gsz[4], goff[4];
  int poff[4];              // offset of group g inside a pooled_g / compact row (sum of c^2 before it)
  int psum;                 // sum of c_g^2
  const double* sim;        // [G][C]
  const int* forced_perm;   // [G][C] or null
  int* perm_out;            // [G][C]
  const float* s_qk;        // [B][C][C]
  const float* rq;          // [B][C]
  const float* rk;          // [B][C]
  const float* temperature; // [4]
  const float* pooled;      // [G][C][C] or null (no cache)
  const float* pooled_g;    // [G][psum]: the cache pooled to every group's c x c
  const float* cw_wt[4];    // conv_width.weight, K-major [c][mp2] (mp2 = 2c rounded up to 32)
  const float* cw_b[4];
  const float* gc_wt[4];    // gating_conv.weight, K-major [c][mp1]
  const float* gc_b[4];
  int n_chunks;             // column chunks of the modulation kernel
  int chunk_g[32], chunk_j0[32];
  int sm_chunks, sm_rc;     // row chunks of the softmax kernel: sm_rc rows of group sm_g[] from row sm_r0[]
  int sm_g[32], sm_r0[32];
  float* att_t;             // [B][Kpad][Mpad]
  float* att_nat;           // [B][C][C]
};

constexpr int GA_JC = 32;   // columns modulated per workgroup
constexpr int GA_LD = 33;   // LDS row stride of the 32-column chunk images
constexpr int GM_THREADS = 1024;   // modulation kernel: one 32-row output tile per wave, up to 16 at once

__device__ __forceinline__ double sort_key(double v) { return v != v ? -INFINITY : v; }

// ---- 3b. the cache pooled to every group's c x c (Inter_CacheModulation.align) ---------------------
// 256 outputs per workgroup; the rows of a window are summed with independent loads in flight (clamped,
// unpredicated), in the order y-major / x-minor of the plain double loop.
__device__ __forceinline__ void pool_groups_block(const GroupAttnArgs& a, int blk, int g) {
  const int idx = blk * 256 + threadIdx.x;
  if (idx >= a.psum) return;
  int gi = 3;
  while (gi > 0 && idx < a.poff[gi]) --gi;
  const int c = a.gsz[gi], C = a.C;
  const int e = idx - a.poff[gi];
  const int r = e / c, s = e - r * c;
  const float* __restrict__ src = a.pooled + (long)g * C * C;
  const int ys = pool_start(r, C, c), ye = pool_end(r, C, c);
  const int xs = pool_start(s, C, c), xe = pool_end(s, C, c);
  double acc = 0.0;
  for (int y = ys; y < ye; ++y) {
    const float* __restrict__ row = src + (long)y * C;
    for (int x0 = xs; x0 < xe; x0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = row[min(x0 + u, xe - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (x0 + u < xe) acc += (double)v[u];
    }
  }
  const_cast<float*>(a.pooled_g)[(long)g * a.psum + idx] = (float)(acc / (double)((ye - ys) * (xe - xs)));
}

// ---- 3a. permutation, gather, scale, row softmax ---------------------------------------------------
// One workgroup per (group, chunk of sm_rc rows, image); every workgroup ranks the channels itself (a
// few microseconds, cheaper than a launch of its own).  Writes the softmax rows in natural orientation;
// without a cache that is already the final attention, so the transposed (GEMM weight) image is written
// too.  Either way the off-diagonal part of the rows kr of the transposed image with kr - off inside
// this chunk is zeroed (the last chunk of the last group also owns the padding rows).
__global__ __launch_bounds__(256) void group_softmax_kernel(const GroupAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ga_smem[];
  // workgroups past the softmax chunks pool the cache for the modulation kernel (independent of the
  // softmax, so it rides in this launch instead of one of its own)
  if ((int)blockIdx.x >= a.sm_chunks) {
    if (blockIdx.y % a.ipg == 0) pool_groups_block(a, blockIdx.x - a.sm_chunks, blockIdx.y / a.ipg);
    return;
  }
  const int gi = a.sm_g[blockIdx.x], r0 = a.sm_r0[blockIdx.x], b = blockIdx.y, tid = threadIdx.x;
  const int C = a.C, c = a.gsz[gi], off = a.goff[gi];
  const int rc = min(a.sm_rc, c - r0);
  const int g = b / a.ipg;
  const int ld = c + 1;
  double* key = reinterpret_cast<double*>(ga_smem);            // [C]
  int* perm = reinterpret_cast<int*>(key + C);                 // [C]
  float* X = reinterpret_cast<float*>(perm + C + (C & 1));     // [sm_rc][ld]

  // ---- channel permutation: descending similarity, ties by ascending channel ----
  if (a.forced_perm) {
    for (int i = tid; i < C; i += 256) perm[i] = a.forced_perm[(long)g * C + i];
  } else {
    for (int i = tid; i < C; i += 256) key[i] = sort_key(a.sim[(long)g * C + i]);
    __syncthreads();
    for (int i = tid; i < C; i += 256) {
      const double ki = key[i];
      int rank = 0;
      for (int j = 0; j < C; ++j) {
        const double kj = key[j];
        rank += (kj > ki) || (kj == ki && j < i);
      }
      perm[rank] = i;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && b % a.ipg == 0)
    for (int i = tid; i < C; i += 256) a.perm_out[(long)g * C + i] = perm[i];

  // ---- gather + scale: X[r][s] = S[perm r][perm s] * rq[perm r] * rk[perm s] * T ----
  const float temp = a.temperature[gi];
  const float* __restrict__ S = a.s_qk + (long)b * C * C;
  // (unrolled: the gathers are dependent-latency bound otherwise, one global round trip per element)
#pragma unroll 8
  for (int e = tid; e < rc * c; e += 256) {
    const int r = e / c, s = e - r * c;
    const int ci = perm[off + r0 + r], cj = perm[off + s];
    X[r * ld + s] = S[(long)ci * C + cj] * a.rq[(long)b * C + ci] * a.rk[(long)b * C + cj] * temp;
  }
  __syncthreads();
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < rc; r += 4) {
      float m = -INFINITY;
      for (int s = lane; s < c; s += 64) m = fmaxf(m, X[r * ld + s]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
      for (int s = lane; s < c; s += 64) {
        const float e = expf(X[r * ld + s] - m);
        X[r * ld + s] = e;
        sum += e;
      }
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.0f / sum;
      for (int s = lane; s < c; s += 64) X[r * ld + s] *= inv;
    }
  }
  __syncthreads();

  const int kpad = ((C + GEMM_KPAD - 1) / GEMM_KPAD) * GEMM_KPAD;
  float* __restrict__ att_t = a.att_t + (long)b * kpad * a.Mpad;
  float* __restrict__ att_nat = a.att_nat + (long)b * C * C;
  for (int e = tid; e < rc * c; e += 256) {
    const int r = e / c, s = e - r * c;
    att_nat[(long)(off + r0 + r) * C + off + s] = X[r * ld + s];
  }
  // zero the part of rows [off + r0, off + r0 + rc) of the transposed image outside the diagonal block
  const bool last = gi == 3 && r0 + rc == c;
  const int row_lo = off + r0, row_hi = last ? kpad : off + r0 + rc;
  for (int e = tid; e < (row_hi - row_lo) * a.Mpad; e += 256) {
    const int kr = row_lo + e / a.Mpad, m = e % a.Mpad;
    const bool inside = kr < off + c && m >= off && m < off + c;
    if (!inside) att_t[(long)kr * a.Mpad + m] = 0.f;
  }
  // softmax row r is column off + r0 + r of the transposed image
  if (a.pooled == nullptr)
    for (int e = tid; e < rc * c; e += 256) {
      const int s = e / rc, r = e - s * rc;
      att_t[(long)(off + s) * a.Mpad + off + r0 + r] = X[r * ld + s];
    }
}

// ---- 3c. inter-cache modulation -----------------------------------------------------------------------
// Conv1d(k=1) mixes ROWS of the c x c attention block, so columns are independent: one workgroup
// per (group, 32-column chunk, image).  With X the softmax block and Pg the pooled cache:
//   Z = Ww (Pg + X) + bw ;  Y = X * Z[:c] + Z[c:] ;  att = Y * gelu(Wg Y + bg) + X
// Both products run on fp32 MFMA (32x32x2): A = K-major packed weights straight from global
// (128-byte rows), B = the chunk image in LDS.
__global__ __launch_bounds__(GM_THREADS) void group_modulate_kernel(const GroupAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
  const int gi = a.chunk_g[blockIdx.x], j0 = a.chunk_j0[blockIdx.x];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int C = a.C, c = a.gsz[gi], off = a.goff[gi];
  const int g = b / a.ipg;
  const int nj = min(GA_JC, c - j0);
  const int mp2 = (2 * c + 31) / 32 * 32, mp1 = (c + 31) / 32 * 32;
  const int ce = (c + 1) & ~1;                     // an MFMA step takes two rows: odd c gets a zero row
  float* Xs = reinterpret_cast<float*>(gm_smem);   // [ce][GA_LD]  softmax chunk
  float* XP = Xs + ce * GA_LD;                     // [ce][GA_LD]  X + pooled cache
  float* Ys = XP + ce * GA_LD;                     // [ce][GA_LD]
  float* Zs = Ys + ce * GA_LD;                     // [mp2][GA_LD]
  float* __restrict__ att_nat = a.att_nat + (long)b * C * C;
  const float* __restrict__ pg = a.pooled_g + (long)g * a.psum + a.poff[gi];

#pragma unroll 4
  for (int e = tid; e < ce * GA_JC; e += GM_THREADS) {
    const int r = e >> 5, jj = e & 31;
    float x = 0.f, xp = 0.f;
    if (jj < nj && r < c) {
      x = att_nat[(long)(off + r) * C + off + j0 + jj];
      xp = x + pg[r * c + j0 + jj];
    }
    Xs[r * GA_LD + jj] = x;
    XP[r * GA_LD + jj] = xp;
  }
  __syncthreads();

  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  // 32 x 32 x c product: weight fragments in explicit batches of GM_KB k-steps, the next batch in flight
  // while the current one is multiplied (left to the compiler the loop is not unrolled and every MFMA
  // waits for its own L2 round trip)
  constexpr int GM_KB = 6;   // group sizes are multiples of 12 for the shipped widths: c / 2 k-steps
  const int nks = (c + 1) >> 1;
  auto tile_product = [&](f32x16_t& acc, const float* __restrict__ ap, const int mp, const float* __restrict__ bp) {
    float av[GM_KB], an[GM_KB];
    const int full = nks / GM_KB * GM_KB;
    if (full > 0) {
#pragma unroll
      for (int i = 0; i < GM_KB; ++i) av[i] = ap[(long)(2 * i) * mp];
    }
    for (int ks = 0; ks < full; ks += GM_KB) {
      const int kn = ks + GM_KB < full ? ks + GM_KB : ks;
#pragma unroll
      for (int i = 0; i < GM_KB; ++i) an[i] = ap[(long)(2 * (kn + i)) * mp];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < GM_KB; ++i)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bp[(2 * (ks + i)) * GA_LD], acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < GM_KB; ++i) av[i] = an[i];
    }
    for (int ks = full; ks < nks; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(long)(2 * ks) * mp], bp[(2 * ks) * GA_LD], acc, 0, 0, 0);
  };
  // Z = Ww XP + bw
  {
    const float* __restrict__ Wt = a.cw_wt[gi];
    const float* __restrict__ bw = a.cw_b[gi];
    for (int mt = wave; mt < mp2 / 32; mt += GM_THREADS / 64) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      tile_product(acc, Wt + (long)half * mp2 + mt * 32 + l31, mp2, XP + half * GA_LD + l31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Zs[row * GA_LD + l31] = acc[r] + (row < 2 * c ? bw[row] : 0.f);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < ce * GA_JC; e += GM_THREADS) {
    const int o = e >> 5, jj = e & 31;
    Ys[o * GA_LD + jj] = (jj < nj && o < c) ? Xs[o * GA_LD + jj] * Zs[o * GA_LD + jj] + Zs[(c + o) * GA_LD + jj] : 0.f;
  }
  __syncthreads();
  // att = Y * gelu(Wg Y + bg) + X
  {
    const float* __restrict__ Wt = a.gc_wt[gi];
    const float* __restrict__ bg = a.gc_b[gi];
    const int kpad = ((C + GEMM_KPAD - 1) / GEMM_KPAD) * GEMM_KPAD;
    float* __restrict__ att_t = a.att_t + (long)b * kpad * a.Mpad;
    for (int mt = wave; mt < mp1 / 32; mt += GM_THREADS / 64) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      tile_product(acc, Wt + (long)half * mp1 + mt * 32 + l31, mp1, Ys + half * GA_LD + l31);
      if (l31 < nj) {
        const int s = j0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (o >= c) continue;
          const float v = Ys[o * GA_LD + l31] * gelu_erf_a(acc[r] + bg[o]) + Xs[o * GA_LD + l31];
          att_nat[(long)(off + o) * C + off + s] = v;
          att_t[(long)(off + s) * a.Mpad + off + o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 4. layer cache + momentum update
// ------------------------------------------------------------------------------------------------
struct CacheUpdArgs {
  int C, ipg;
  int gsz[4], goff[4];
  const float* att_nat;  // [B][C][C]
  const float* pooled;   // [G][C][C] or null
  float* cache_out;      // [G][C][C]
};

__global__ __launch_bounds__(256) void cache_update_kernel(const CacheUpdArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  const int C = a.C;
  if (idx >= C * C) return;
  const int oy = idx / C, ox = idx - oy * C;
  float total = 0.f;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int c = a.gsz[gi], off = a.goff[gi];
    // bilinear, align_corners = False: src = scale * (dst + 0.5) - 0.5 clamped at 0
    const float scale = (float)c / (float)C;
    float sy = scale * ((float)oy + 0.5f) - 0.5f;
    float sx = scale * ((float)ox + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = min(y0, c - 1);
    x0 = min(x0, c - 1);
    const int y1 = min(y0 + 1, c - 1), x1 = min(x0 + 1, c - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    for (int m = 0; m < a.ipg; ++m) {
      const float* __restrict__ A = a.att_nat + ((long)(g * a.ipg + m) * C + off) * C + off;
      v00 += A[(long)y0 * C + x0];
      v01 += A[(long)y0 * C + x1];
      v10 += A[(long)y1 * C + x0];
      v11 += A[(long)y1 * C + x1];
    }
    if (a.ipg > 1) {
      const float inv = (float)a.ipg;
      v00 /= inv; v01 /= inv; v10 /= inv; v11 /= inv;
    }
    const float up = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    total += floorf((up * (float)c) / (float)C);
  }
  float outv;
  if (a.pooled)
    outv = a.pooled[(long)g * C * C + idx] * 0.9f + total * 0.1f;
  else
    outv = total * 0.9f;
  a.cache_out[(long)g * C * C + idx] = outv;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// enough workgroups to fill the chip, but at least 2 chunks (64 px) per split
static int stats_splits(int tiles, int B, int P) {
  const int chunks = (P + ST_CH - 1) / ST_CH;
  // about 512 workgroups (one resident round of two per CU): a sweep over 128 ... 4096 at batches 1 to 32 had its
  // minimum there (batch 32: statistics + reduction 12.3 ms per step against 12.6 at 1024 and 14.2 at 256)
  const int want = (512 + tiles * tiles * B - 1) / (tiles * tiles * B);
  // more slabs cost the reduction kernel time in proportion to splits * B: allow many only while
  // the batch is small
  // (a single large image, e.g. 1280 x 720: 512 slabs, so that every CU holds two workgroups as it does in a batch;
  // measured slower for 256 x 256 and 640 x 360 planes, where the slab reduction grows by what the statistics gain)
  const int cap = std::min(chunks >= 16384 ? 512 : 256, std::max(16, 512 / (tiles * tiles * B)));
  // a workgroup is latency-bound per chunk (one workgroup per CU, register double buffer only), so a
  // small problem is cut down to 2 chunks (64 px) per split: measured faster than 4 or 8 at B = 1
  return std::min(cap, std::max(1, std::min(want, chunks / 2)));
}

HintAttnShape hint_attn_shape(int B, int C, int P, int batch_mode) {
  HintAttnShape sh{};
  sh.B = B; sh.C = C; sh.P = P;
  sh.G = batch_mode == JIRF_BATCH_COUPLED ? 1 : B;
  const int ratio[4] = {1, 2, 2, 3};
  int off = 0;
  for (int i = 0; i < 4; ++i) {
    // int(ratio / 8 * C) evaluated like the reference: double arithmetic, truncation
    sh.gsz[i] = (int)((double)ratio[i] / 8.0 * (double)C);
    sh.goff[i] = off;
    off += sh.gsz[i];
  }
  sh.splits = stats_splits((C + SQ_T - 1) / SQ_T, B, P);
  return sh;
}

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

void pack_hint_modulator(const float* w, int M, int K, std::vector<float>& out) {
  const int mp = (M + 31) / 32 * 32;
  out.assign((size_t)((K + 1) & ~1) * mp, 0.f);   // an MFMA step takes two k rows: odd K gets a zero row
  for (int o = 0; o < M; ++o)
    for (int k = 0; k < K; ++k) out[(size_t)k * mp + o] = w[(size_t)o * K + k];
}

size_t hint_attn_scratch_carve(const HintAttnShape& sh, void* base, HintAttnScratch* o) {
  size_t off = 0;
  const size_t C = sh.C, B = sh.B, G = sh.G, S = sh.splits;
  auto take = [&](size_t bytes) {
    void* p = base ? static_cast<unsigned char*>(base) + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const size_t kpad = (C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const size_t mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
  o->gram_part = static_cast<double*>(take(G * S * C * C * 8));
  o->qk_part = static_cast<float*>(take(B * S * C * C * 4));
  o->qsum_part = static_cast<double*>(take(G * S * C * 8));
  o->q2_part = static_cast<float*>(take(B * S * C * 4));
  o->k2_part = static_cast<float*>(take(B * S * C * 4));
  o->qmean = static_cast<float*>(take(G == B ? 0 : C * (size_t)sh.P * 4));
  o->sim = static_cast<double*>(take(G * C * 8));
  o->s_qk = static_cast<float*>(take(B * C * C * 4));
  o->rq = static_cast<float*>(take(B * C * 4));
  o->rk = static_cast<float*>(take(B * C * 4));
  o->att_t = static_cast<float*>(take(B * kpad * mpad * 4));
  o->att3 = static_cast<uint16_t*>(take(B * 3 * kpad * mpad * 2));
  o->att_nat = static_cast<float*>(take(B * C * C * 4));
  o->pooled = static_cast<float*>(take(G * C * C * 4));
  o->pooled_g = static_cast<float*>(take(G * C * C * 4));  // sum of c_g^2 <= C^2
  return off;
}

void launch_hint_attention(const HintAttnShape& sh, TV q, TV k, const HintAttnWeights& w,
                           const float* cache_in, int cache_c, float* cache_out,
                           const HintAttnScratch& sc, int* perm_out, const int* forced_perm,
                           hipStream_t s) {
  const int B = sh.B, C = sh.C, P = sh.P, G = sh.G, ipg = B / G;
  JIRF_REQUIRE(sh.gsz[0] > 0 && sh.goff[3] + sh.gsz[3] <= C, JIRF_EINVAL,
               "hint attention: channel count %d too small for 4 groups", C);
  JIRF_REQUIRE(P >= 2, JIRF_EINVAL, "hint attention: needs at least 2 pixels");
  const int tiles = (C + SQ_T - 1) / SQ_T;
  const int chunks = (P + ST_CH - 1) / ST_CH;
  const int cps = (chunks + sh.splits - 1) / sh.splits;
  float* q2_part = sc.q2_part;
  float* k2_part = sc.k2_part;
  double* sim = sc.sim;

  StatsArgs st{};
  st.q = q.p; st.q_bs = q.bs; st.k = k.p; st.k_bs = k.bs;
  st.C = C; st.P = P; st.tiles = tiles; st.splits = sh.splits; st.chunks_per_split = cps;
  st.qk_part = sc.qk_part; st.gram_part = sc.gram_part; st.qsum_part = sc.qsum_part;
  st.q2_part = q2_part; st.k2_part = k2_part;
  const double tile_px = (double)C * C * (double)P;
  if (ipg == 1) {
    st.qg = q.p; st.qg_bs = q.bs;
    Scope scope(intern_label("hint_chan_stats C%d P%d S%d", C, P, sh.splits), 4.0 * tile_px * B, 8.0 * B * (double)C * P, s);
    launch_chan_stats96<true, true>(st, B, s);
  } else {
    {
      Scope scope("hint_batch_mean", (double)B * C * P, 4.0 * (B + 1) * (double)C * P, s);
      const long n = (long)C * P;
      hipLaunchKernelGGL(batch_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q.p,
                         q.bs, sc.qmean, B, n);
    }
    {
      Scope scope("hint_chan_stats", 2.0 * tile_px * B, 8.0 * B * (double)C * P, s);
      launch_chan_stats96<true, false>(st, B, s);
    }
    st.qg = sc.qmean; st.qg_bs = 0;
    Scope scope("hint_chan_stats", 2.0 * tile_px, 4.0 * (double)C * P, s);
    launch_chan_stats96<false, true>(st, 1, s);
  }
  JIRF_HIP_CHECK(hipGetLastError());
  {
    Scope scope(intern_label("hint_regroup_rows C%d", C), 3.0 * C * (double)C * sh.splits * B,
                (double)C * C * sh.splits * (4.0 * B + 8.0 * G), s);
    const size_t smem = 2 * (size_t)C * sizeof(double);
    launch_reduce_slabs(sc.qk_part, sc.gram_part, sc.qsum_part, q2_part, k2_part, C, sh.splits, B, G, s);
    hipLaunchKernelGGL(regroup_rows_kernel, dim3((C + RG_ROWS - 1) / RG_ROWS, B), dim3(256), smem, s, sc.qk_part, sc.gram_part,
                       sc.qsum_part, q2_part, k2_part, sc.s_qk, sc.rq, sc.rk, sim, C, P, sh.splits, 1,
                       ipg, 0.f);
    JIRF_HIP_CHECK(hipGetLastError());
  }
  const float* pooled = nullptr;
  if (cache_in && cache_c == C) {
    pooled = cache_in;   // AdaptiveAvgPool2d to the same size is the identity: no copy
  } else if (cache_in) {
    Scope scope("hint_cache_pool", 0.0, 4.0 * G * ((double)cache_c * cache_c + (double)C * C), s);
    hipLaunchKernelGGL(cache_pool_kernel, dim3((C * C + 255) / 256, G), dim3(256), 0, s, cache_in,
                       cache_c, sc.pooled, C);
    JIRF_HIP_CHECK(hipGetLastError());
    pooled = sc.pooled;
  }
  {
    GroupAttnArgs ga{};
    ga.C = C; ga.G = G; ga.ipg = ipg;
    ga.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
    int cmax = 0, psum = 0, nch = 0;
    for (int i = 0; i < 4; ++i) {
      ga.gsz[i] = sh.gsz[i]; ga.goff[i] = sh.goff[i];
      ga.cw_wt[i] = w.cw_wt[i]; ga.cw_b[i] = w.cw_b[i]; ga.gc_wt[i] = w.gc_wt[i]; ga.gc_b[i] = w.gc_b[i];
      cmax = std::max(cmax, sh.gsz[i]);
      ga.poff[i] = psum;
      psum += sh.gsz[i] * sh.gsz[i];
      for (int j0 = 0; j0 < sh.gsz[i]; j0 += GA_JC) {
        JIRF_REQUIRE(nch < 32, JIRF_EINVAL, "hint attention: C=%d needs more than 32 column chunks", C);
        ga.chunk_g[nch] = i;
        ga.chunk_j0[nch] = j0;
        ++nch;
      }
    }
    ga.psum = psum; ga.n_chunks = nch;
    // softmax row chunks: 16 rows each unless that needs more than 32 workgroups per image
    int rcs = 16;
    auto sm_count = [&](int rc) { int n = 0; for (int i = 0; i < 4; ++i) n += (sh.gsz[i] + rc - 1) / rc; return n; };
    while (sm_count(rcs) > 32) rcs *= 2;
    ga.sm_rc = rcs; ga.sm_chunks = 0;
    for (int i = 0; i < 4; ++i)
      for (int r0 = 0; r0 < sh.gsz[i]; r0 += rcs) {
        ga.sm_g[ga.sm_chunks] = i;
        ga.sm_r0[ga.sm_chunks] = r0;
        ++ga.sm_chunks;
      }
    ga.sim = sim; ga.forced_perm = forced_perm; ga.perm_out = perm_out;
    ga.s_qk = sc.s_qk; ga.rq = sc.rq; ga.rk = sc.rk; ga.temperature = w.temperature;
    ga.pooled = pooled; ga.pooled_g = sc.pooled_g; ga.att_t = sc.att_t; ga.att_nat = sc.att_nat;
    const size_t smem_a = (size_t)C * 8 + ((size_t)C + (C & 1)) * 4 + (size_t)std::min(rcs, cmax) * (cmax + 1) * 4;
    const size_t smem_c = ((size_t)3 * ((cmax + 1) & ~1) + (2 * cmax + 31) / 32 * 32) * GA_LD * 4;
    JIRF_REQUIRE(smem_a <= 160 * 1024 && smem_c <= 160 * 1024, JIRF_EINVAL,
                 "hint attention: channel group of %d does not fit LDS (%zu / %zu bytes)", cmax, smem_a, smem_c);
    static DynLdsOnce once_softmax, once_modulate;
    allow_dynamic_lds(once_softmax, reinterpret_cast<const void*>(group_softmax_kernel), 160 * 1024);
    allow_dynamic_lds(once_modulate, reinterpret_cast<const void*>(group_modulate_kernel), 160 * 1024);
    {
      // with a cache: (psum + 255) / 256 more workgroups per image pool it to the groups' sizes
      const int pool_blocks = pooled ? (psum + 255) / 256 : 0;
      Scope scope(intern_label("hint_group_softmax C%d", C), 0.0,
                  12.0 * B * (double)C * C + (pooled ? 4.0 * G * ((double)C * C + psum) : 0.0), s);
      hipLaunchKernelGGL(group_softmax_kernel, dim3(ga.sm_chunks + pool_blocks, B), dim3(256), smem_a, s, ga);
      JIRF_HIP_CHECK(hipGetLastError());
    }
    if (pooled) {
      double fl = 0.0;
      for (int i = 0; i < 4; ++i) fl += 6.0 * sh.gsz[i] * (double)sh.gsz[i] * sh.gsz[i];
      Scope scope(intern_label("hint_group_modulate C%d", C), fl * B, 16.0 * B * (double)psum, s);
      hipLaunchKernelGGL(group_modulate_kernel, dim3(nch, B), dim3(GM_THREADS), smem_c, s, ga);
      JIRF_HIP_CHECK(hipGetLastError());
    }
  }
  {
    CacheUpdArgs cu{};
    cu.C = C; cu.ipg = ipg;
    for (int i = 0; i < 4; ++i) { cu.gsz[i] = sh.gsz[i]; cu.goff[i] = sh.goff[i]; }
    cu.att_nat = sc.att_nat; cu.pooled = pooled; cu.cache_out = cache_out;
    Scope scope(intern_label("hint_cache_update C%d", C), 40.0 * G * (double)C * C, 12.0 * G * (double)C * C, s);
    hipLaunchKernelGGL(cache_update_kernel, dim3((C * C + 255) / 256, G), dim3(256), 0, s, cu);
    JIRF_HIP_CHECK(hipGetLastError());
  }
}

// ------------------------------------------------------------------------------------------------
// Head-wise channel attention of ASTv2 (reference basicsr/models/archs/astv2_arch.py:158-206):
// fixed heads of ch = C / heads channels, no regrouping, no cache.  One workgroup per (head, image):
//   A  = (q^ k^T) * T_head          S0 = softmax_rows(A)          R = relu(A)^2 ; R = gelu(R) * R
//   Z[o][j] = sum_i Wca[o][i] R[i][j] (+ b[o])                    (1x1 conv over the ROW axis)
//   A'[r][c] = S0[r][c] * (1 + Z[c][r]) + Z[ch + c][r]            (scale / shift are Z's halves,
//                                                                   transposed by the permutes)
// ------------------------------------------------------------------------------------------------
struct HeadAttnArgs {
  int C, heads, ch, Mpad;
  const float* s_qk;         // [B][C][C]
  const float* rq;           // [B][C]
  const float* rk;           // [B][C]
  const float* temperature;  // [heads]
  const float* ca_w;         // [2ch][ch]
  const float* ca_b;         // [2ch] or null
  float* att_t;              // [B][Kpad][Mpad]
};

__global__ __launch_bounds__(256) void head_attn_kernel(const HeadAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ha_smem[];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C = a.C, c = a.ch, off = h * c, ld = c + 1;
  float* S0 = reinterpret_cast<float*>(ha_smem);  // [c][ld] softmax branch
  float* R = S0 + c * ld;                         // [c][ld] sparse branch
  float* Z = R + c * ld;                          // [2c][ld]
  const float temp = a.temperature[h];
  const float* __restrict__ S = a.s_qk + (long)b * C * C;
  for (int e = tid; e < c * c; e += 256) {
    const int r = e / c, s = e - r * c;
    const float v = S[(long)(off + r) * C + off + s] * a.rq[(long)b * C + off + r] *
                    a.rk[(long)b * C + off + s] * temp;
    S0[r * ld + s] = v;
    if (a.ca_w) {
      const float p = fmaxf(v, 0.f);
      const float p2 = p * p;
      R[r * ld + s] = gelu_erf_a(p2) * p2;
    }
  }
  __syncthreads();
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < c; r += 4) {
      float m = -INFINITY;
      for (int s = lane; s < c; s += 64) m = fmaxf(m, S0[r * ld + s]);
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
      for (int s = lane; s < c; s += 64) {
        const float e = expf(S0[r * ld + s] - m);
        S0[r * ld + s] = e;
        sum += e;
      }
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.0f / sum;
      for (int s = lane; s < c; s += 64) S0[r * ld + s] *= inv;
    }
  }
  for (int e = tid; a.ca_w && e < 2 * c * c; e += 256) {
    const int o = e / c, j = e - o * c;
    float z = a.ca_b ? a.ca_b[o] : 0.f;
    const float* __restrict__ w = a.ca_w + (long)o * c;
    for (int i = 0; i < c; ++i) z = fmaf(w[i], R[i * ld + j], z);
    Z[o * ld + j] = z;
  }
  __syncthreads();
  const int kpad = ((C + GEMM_KPAD - 1) / GEMM_KPAD) * GEMM_KPAD;
  float* __restrict__ att_t = a.att_t + (long)b * kpad * a.Mpad;
  // rows [off, off + c) of the transposed image are this head's (the last head also owns the
  // padding rows); everything outside the diagonal block must read as zero in the GEMM
  const int row_end = h == a.heads - 1 ? kpad : off + c;
  for (int e = tid; e < (row_end - off) * a.Mpad; e += 256) {
    const int kr = off + e / a.Mpad, m = e % a.Mpad;
    float v = 0.f;
    if (kr < off + c && m >= off && m < off + c) {
      const int col = kr - off, r = m - off;   // att_t[k = col][m = row] = A'[row][col]
      v = a.ca_w ? S0[r * ld + col] * (1.0f + Z[col * ld + r]) + Z[(c + col) * ld + r] : S0[r * ld + col];
    }
    att_t[(long)kr * a.Mpad + m] = v;
  }
}

size_t head_attn_scratch_carve(int B, int C, int P, int splits, void* base, HeadAttnScratch* o) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base ? static_cast<unsigned char*>(base) + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const size_t kpad = ((size_t)C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const size_t mpad = ((size_t)C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
  o->qk_part = static_cast<float*>(take((size_t)B * splits * C * C * 4));
  o->q2_part = static_cast<float*>(take((size_t)B * splits * C * 4));
  o->k2_part = static_cast<float*>(take((size_t)B * splits * C * 4));
  o->s_qk = static_cast<float*>(take((size_t)B * C * C * 4));
  o->rq = static_cast<float*>(take((size_t)B * C * 4));
  o->rk = static_cast<float*>(take((size_t)B * C * 4));
  o->att_t = static_cast<float*>(take((size_t)B * kpad * mpad * 4));
  o->att3 = static_cast<uint16_t*>(take((size_t)B * 3 * kpad * mpad * 2));
  return off;
}

// heads whose size divides the 96-channel super tile use that kernel (diagonal tile pairs only)
static bool head_uses_sq96(int ch) { return SQ_T % ch == 0; }
// upper bound over both statistics kernels (sizes the partial-sum scratch)
int head_attn_splits(int B, int C, int P) {
  return std::max(stats_splits((C + ST_T - 1) / ST_T, B, P), stats_splits((C + SQ_T - 1) / SQ_T, B, P));
}

void launch_head_attention(int B, int C, int P, int heads, TV q, TV k, const float* temperature,
                           const float* ca_w, const float* ca_b, const HeadAttnScratch& sc,
                           hipStream_t s, float norm_add_eps) {
  JIRF_REQUIRE(heads > 0 && C % heads == 0, JIRF_EINVAL, "head attention: C=%d is not divisible by heads=%d", C, heads);
  const int ch = C / heads;
  const size_t smem = (size_t)4 * ch * (ch + 1) * 4;
  JIRF_REQUIRE(smem <= 160 * 1024, JIRF_EINVAL, "head attention: %d channels per head do not fit LDS", ch);
  const bool sq96 = head_uses_sq96(ch);
  const int tiles = sq96 ? (C + SQ_T - 1) / SQ_T : (C + ST_T - 1) / ST_T;
  const int splits = stats_splits(tiles, B, P);
  const int chunks = (P + ST_CH - 1) / ST_CH;
  StatsArgs st{};
  st.q = q.p; st.q_bs = q.bs; st.k = k.p; st.k_bs = k.bs;
  st.C = C; st.P = P; st.tiles = tiles; st.splits = splits;
  st.chunks_per_split = (chunks + splits - 1) / splits;
  st.head_ch = ch;
  st.qk_part = sc.qk_part; st.q2_part = sc.q2_part; st.k2_part = sc.k2_part;
  {
    Scope scope(intern_label("head_chan_stats C%d P%d S%d", C, P, splits), 2.0 * C * (double)ch * P * B,
                8.0 * B * (double)C * P, s);
    if (sq96) launch_chan_stats96<true, false>(st, B, s);
    else launch_chan_stats<true, false>(st, B, s);
  }
  {
    Scope scope("head_reduce_rows", 1.0 * C * (double)C * splits * B, 4.0 * C * (double)C * splits * B, s);
    launch_reduce_slabs(sc.qk_part, nullptr, nullptr, sc.q2_part, sc.k2_part, C, splits, B, 0, s);
    hipLaunchKernelGGL(regroup_rows_kernel, dim3((C + RG_ROWS - 1) / RG_ROWS, B), dim3(256), 0, s, sc.qk_part, nullptr, nullptr,
                       sc.q2_part, sc.k2_part, sc.s_qk, sc.rq, sc.rk, nullptr, C, P, splits, 1, 1, norm_add_eps);
    JIRF_HIP_CHECK(hipGetLastError());
  }
  {
    HeadAttnArgs ha{};
    ha.C = C; ha.heads = heads; ha.ch = ch;
    ha.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
    ha.s_qk = sc.s_qk; ha.rq = sc.rq; ha.rk = sc.rk; ha.temperature = temperature;
    ha.ca_w = ca_w; ha.ca_b = ca_b; ha.att_t = sc.att_t;
    static DynLdsOnce once;
    allow_dynamic_lds(once, reinterpret_cast<const void*>(head_attn_kernel), 160 * 1024);
    Scope scope(intern_label("head_attn C%d ch%d", C, ch), 4.0 * ch * (double)ch * ch * heads * B,
                8.0 * B * (double)C * C, s);
    hipLaunchKernelGGL(head_attn_kernel, dim3(heads, B), dim3(256), smem, s, ha);
    JIRF_HIP_CHECK(hipGetLastError());
  }
}

}  // namespace jirf
