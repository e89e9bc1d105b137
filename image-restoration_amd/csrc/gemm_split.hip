// Split-bf16 GEMM with the pixel operand held in registers (gfx950), for short reductions (K <= 96).
//
// Every fp32 operand is written exactly as hi + mid + lo with three bf16 values (8 significand bits each,
// 24 together); a product a * b is then the sum of nine exact bf16 x bf16 products, of which the six largest
// are accumulated in fp32 on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate):
//   hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid      (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a b|)
// so the result differs from an fp32 fma chain by rounding of the same order as a different summation order.
//
// Layout.  A wave owns 32 pixels for its whole life.  Lane (pixel l31, half h) loads the channels
// 16 s + 8 h + i (i < 8) of its pixel with one dword load each — 128-byte row segments, the row advance in
// the scalar offset — so all channels of a pixel sit in lanes l and l + 32: LayerNorm is two register passes
// and one lane swap, and the normalised values are split and packed straight into MFMA B fragments (eight
// consecutive k per lane).  The activation tile never passes through LDS.  The weights stream through LDS
// instead: one chunk = the three bf16 planes of a 32-row output tile, copied by the whole workgroup a tile
// ahead (global memory -> LDS with no register in between, one barrier per tile) and read by every wave as A
// fragments, 16 bytes per lane, conflict-free.  Each output tile is 36 MFMAs at K = 96 and 16 dword stores
// per lane.  125 registers at K = 96: four waves per SIMD, four workgroups (of 128 pixels) per CU.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "split_bf16.h"

namespace jirf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// t[l] + t[l ^ 32]: v_permlane32_swap exchanges the upper half of its first operand with the lower half of
// its second; fed two copies of t, each lane then holds its own value in one and its partner's in the other.
// (Written as assembly: through the builtin with the same value twice, the compiler takes the two results
// for equal and adds one of them to itself.)
__device__ __forceinline__ float swap32_sum(float t) {
  float u = t;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(u));
  return t + u;
}

// One 1 KiB piece of a weight chunk, global memory -> LDS with no register in between: lane l's 16 bytes land at
// lds_dst + 16 l.  Written as assembly so that the compiler does not count it: it would otherwise wait for the
// copy before the next LDS read of the OTHER buffer.  The issuing wave waits (vmcnt) before the barrier that
// hands the buffer to the readers.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Waves per workgroup (32 pixels each): the waves of a workgroup share every weight chunk, so eight of them halve
// the chunk traffic from L2 of four (LayerNorm GEMM 96 -> 510 at 65 536 pixels x 32 images: 1.41 -> 1.27 ms; sixteen:
// no further gain, two: 1.77); grids too small to give every CU a few workgroups of eight use four.
static int b3_waves(int P, int B) { return (long)((P + 255) / 256) * B >= 1024 ? 8 : 4; }

template <int NS, bool LN, int B3_WAVES, bool GELU>
__global__ __launch_bounds__(64 * B3_WAVES, 4) void gemm_b3_kernel(const GemmArgs a, const int tiles_n) {
  constexpr int B3_BN = 32 * B3_WAVES;          // pixels per workgroup
  constexpr int K = NS * 16;
  constexpr int NG = NS * 2;                    // groups of eight k
  constexpr int CH = 3 * NG * 32;               // 16-byte slots of one weight chunk (plane, k group, row)
  constexpr int PIECES = CH / 64;               // 1 KiB pieces of a chunk
  constexpr int PW = (PIECES + B3_WAVES - 1) / B3_WAVES;   // pieces a wave copies
  constexpr int NT = 64 * B3_WAVES;
  __shared__ __attribute__((aligned(16))) u32x4 Abuf[2][CH];
  extern __shared__ __attribute__((aligned(16))) float b3_bias[];   // [ceil32(M)]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x - b * tiles_n) * B3_BN;
  const int P = a.P, M = a.M;
  const int n_mt = (M + 31) >> 5;

  // ---- weight chunks: [plane][k group][Mpad] 16-byte words in global memory; wave w copies pieces w, w + 4, ...
  const u32x4* __restrict__ W3 = reinterpret_cast<const u32x4*>(a.wt3);
  const int plane16 = (int)(a.wt3_plane / 8);
  const u32x4* w_src[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int q = min((wave + i * B3_WAVES) * 64 + lane, CH - 1);
    const int pl = q / (NG * 32), g = (q >> 5) % NG, m = q & 31;
    w_src[i] = W3 + pl * plane16 + g * a.Mpad + m;
  }
  const unsigned abuf0 = (unsigned)(uintptr_t)&Abuf[0][0];   // LDS byte address
  auto copy_chunk = [&](int mt, int buf) {
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (PIECES % B3_WAVES == 0 || wave + i * B3_WAVES < PIECES)   // wave-uniform
        dma16(w_src[i] + mt * 32, abuf0 + (unsigned)(buf * CH + (wave + i * B3_WAVES) * 64) * 16u);
  };
  copy_chunk(0, 0);
  if (n_mt > 1) copy_chunk(1, 1);

  // ---- the wave's 32 pixels: load, normalise, split ---------------------------------------------------
  const float* __restrict__ xb = a.x.p + (long)b * a.x.bs;
  const int pix = n0 + wave * 32 + l31;
  const int x_voff = (8 * half * P + min(pix, P - 1)) * 4;
  // (plain loads may take their first k_split rows, a multiple of 8, from a second tensor of the same arena - a
  //  concatenation never made: the same descriptor, those lanes' offsets moved by the distance between the two)
  const long x2_delta = !LN && a.k_split > 0 ? (long)((a.x2.p + (long)b * a.x2.bs) - xb) * 4 : 0;
  const __amdgpu_buffer_rsrc_t xs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (int)min((long)K * P * 4 + x2_delta, 0x7fffffffL), 0x00020000);
  float raw[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int vo = x_voff + (16 * s + 8 * half < a.k_split && !LN ? (int)x2_delta : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      raw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xs_rsrc, vo, (16 * s + i) * P * 4, 0));
  }

  {   // the output bias waits in LDS (rows past M as zeros)
    const float* __restrict__ bsrc = a.bias ? a.bias + (long)b * a.bias_bs : nullptr;
    for (int m = tid; m < n_mt * 32; m += NT) b3_bias[m] = bsrc && m < M ? bsrc[m] : 0.f;
  }

  if constexpr (LN) {
    // mean and population variance in two passes, as the reference's mean / var(unbiased=False)
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += raw[s][i];
    sum = swap32_sum(sum);
    const float inv_k = 1.0f / (float)K;
    const float mu = sum * inv_k;
    float q = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = raw[s][i] - mu;
        q = fmaf(d, d, q);
      }
    q = swap32_sum(q);
    const float rs = __builtin_amdgcn_rsqf(q * inv_k + 1e-5f);
    const float mc = a.ln_center ? mu : 0.f;   // BiasFree does not centre
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[s][i] = (raw[s][i] - mc) * rs;
  }
  bf16x8 Bh[NS], Bm[NS], Bl[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) split8(raw[s], Bh[s], Bm[s], Bl[s]);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunks 0 and 1 are in LDS
  __syncthreads();

  // ---- output tiles ------------------------------------------------------------------------------------
  // Per tile: MFMAs, wait for the copy of the next chunk, barrier, request the chunk after next into the buffer
  // just read, and only then the stores.  The CU's vector-memory path is one queue: a copy requested after a
  // tile's stores waits behind them (microseconds when the chip writes at full rate); requested before them it
  // has the whole next tile to land.
  float* __restrict__ yb = a.y.p + (long)b * a.y.bs;
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
  const int OOB = 0x7fffffff;
  const int y_voff = pix < P ? (4 * half * P + pix) * 4 : OOB;

  for (int mt = 0; mt < n_mt; ++mt) {
    const int buf = mt & 1;
    const int m0 = mt * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b3_bias[m0 + (r & 3) + 8 * (r >> 2) + 4 * half];
    const bf16x8* __restrict__ Ab = reinterpret_cast<const bf16x8*>(&Abuf[buf][half * 32 + l31]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf16x8 ah = Ab[(0 * NG + 2 * s) * 32];
      const bf16x8 am = Ab[(1 * NG + 2 * s) * 32];
      const bf16x8 al = Ab[(2 * NG + 2 * s) * 32];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, Bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, Bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, Bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bh[s], acc, 0, 0, 0);
    }
    // the next chunk has landed (and, the counter being one, the previous tile's stores are done)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mt + 2 < n_mt) copy_chunk(mt + 2, buf);
    if (m0 + 32 <= M) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(GELU ? gelu_fast(acc[r]) : acc[r]), y_rsrc, y_voff,
                                              (m0 + (r & 3) + 8 * (r >> 2)) * P * 4, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2);
        const int vo = row + 4 * half < M ? y_voff : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(GELU ? gelu_fast(acc[r]) : acc[r]), y_rsrc, vo,
                                              min(row, M - 1) * P * 4, 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same idea turned round for short OUTPUTS (M <= 96, any reduction length: the project_out / gate GEMMs, whose
// input is several times wider than their output).  A wave still owns 32 pixels, but now keeps the accumulators
// of all its output rows (up to three 32-row tiles) and walks over the reduction in chunks of 32: the chunk's 16
// channels per lane are loaded a chunk ahead, split on arrival, and meet the chunk's weights (every output row x
// 32 k x three planes, at most 18 KiB) in LDS.  No store happens inside the loop, so the loads' wait costs nothing
// extra; bias, residual or gate, and the stores come once at the end.
// ------------------------------------------------------------------------------------------------------------
template <int NT, int EPI, int B3_WAVES, int LOADER>
__global__ __launch_bounds__(64 * B3_WAVES, B3_WAVES == 8 ? 4 : 3) void gemm_k3_kernel(const GemmArgs a, const int tiles_n) {
  constexpr bool CONV = LOADER == LD_CONV3;
  constexpr int B3_BN = 32 * B3_WAVES;          // pixels per workgroup
  constexpr int R = NT * 32;                    // weight rows per k group
  constexpr int CH = 3 * 4 * R;                 // 16-byte slots of one chunk (plane, k group, row)
  constexpr int PIECES = CH / 64;
  constexpr int PW = (PIECES + B3_WAVES - 1) / B3_WAVES;
  __shared__ __attribute__((aligned(16))) u32x4 Abuf[2][CH];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x - b * tiles_n) * B3_BN;
  const int P = a.P, M = a.M, K = a.K;
  const int nch = (K + 31) >> 5;

  const u32x4* __restrict__ W3 = reinterpret_cast<const u32x4*>(a.wt3) + (long)b * (a.wt3_bs / 8);
  const int plane16 = (int)(a.wt3_plane / 8);
  const int ngroups = (K + GEMM_KPAD - 1) / GEMM_KPAD * 2;     // k groups of eight in a plane
  int w_off[PW], w_g[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int q = min((wave + i * B3_WAVES) * 64 + lane, CH - 1);
    const int pl = q / (4 * R), m = q % R;
    w_g[i] = (q / R) & 3;
    w_off[i] = pl * plane16 + min(m, a.Mpad - 1);
  }
  const unsigned abuf0 = (unsigned)(uintptr_t)&Abuf[0][0];
  // (a reduction that is no multiple of 32 ends in half a chunk: its missing k groups re-read the last one, and meet
  //  zeros on the activation side)
  auto copy_chunk = [&](int kc, int buf) {
#pragma unroll
    for (int i = 0; i < PW; ++i)
      if (PIECES % B3_WAVES == 0 || wave + i * B3_WAVES < PIECES)   // wave-uniform
        dma16(W3 + w_off[i] + (long)min(4 * kc + w_g[i], ngroups - 1) * a.Mpad,
              abuf0 + (unsigned)(buf * CH + (wave + i * B3_WAVES) * 64) * 16u);
  };

  const float* __restrict__ xb = a.x.p + (long)b * a.x.bs;
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, (int)min((long)K * P * 4, 0x7fffffffL), 0x00020000);
  const int pix = n0 + wave * 32 + l31;
  const int OOB = 0x7fffffff;
  const int x_voff = (8 * half * P + min(pix, P - 1)) * 4;
  // implicit k x k convolution (k = tap * Cin + c, Cin % 16 == 0): the sixteen k of a step share one tap, a scalar;
  // the lane's input coordinate moves with it, the plane advance is the scalar offset, and a tap in the zero padding
  // is an out-of-range offset
  int cv_y = 0, cv_x = 0;
  __amdgpu_buffer_rsrc_t cv_rsrc = x_rsrc;
  if constexpr (CONV) {
    const int pc = min(pix, P - 1), yy = pc / a.W;
    cv_y = yy * a.stride - a.pad;
    cv_x = (pc - yy * a.W) * a.stride - a.pad;
    cv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0,
                                                (int)min((long)a.Cin * a.Hin * a.Win * 4, 0x7fffffffL), 0x00020000);
  }
  float raw[2][8];
  auto fetch = [&](int kc) {   // channels 32 kc + 16 s + 8 half + i; rows past K read as zero (out-of-range offset)
    if constexpr (CONV) {
      const int plane_b = a.Hin * a.Win * 4;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int k0 = 32 * kc + 16 * s;
        const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
        const int ky = tap / a.ks, kx = tap - ky * a.ks;
        const int iy = cv_y + ky, ix = cv_x + kx;
        const bool ok = k0 < K && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        const int vo = ok ? (iy * a.Win + ix) * 4 + 8 * half * plane_b : OOB;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          raw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cv_rsrc, vo, min(c0 + i, a.Cin - 1) * plane_b, 0));
      }
      return;
    }
    if constexpr (LOADER == LD_GATHER) {   // row kidx[k] instead of row k (the same order for every image of a group)
      const int* __restrict__ kx = a.kidx + (long)(b / a.imgs_per_group) * K;
      const int pc4 = min(pix, P - 1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = 32 * kc + 16 * s + i;
          const int r0 = kx[min(k, K - 1)], r1 = kx[min(k + 8, K - 1)];
          const int vo = k + 8 * half < K ? ((half ? r1 : r0) * P + pc4) * 4 : OOB;
          raw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, vo, 0, 0));
        }
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 32 * kc + 16 * s + i;      // (+ 8 half in the lane part)
        const int vo = k + 8 * half < K ? x_voff : OOB;
        raw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, vo, min(k, K - 1) * P * 4, 0));
      }
  };
  copy_chunk(0, 0);
  fetch(0);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  bf16x8 Bh[2], Bm[2], Bl[2];
  split8(raw[0], Bh[0], Bm[0], Bl[0]);           // (the compiler waits for the loads here)
  split8(raw[1], Bh[1], Bm[1], Bl[1]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk 0 of the weights is in LDS
  __syncthreads();

  for (int kc = 0; kc < nch; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nch) {
      copy_chunk(kc + 1, buf ^ 1);
      fetch(kc + 1);
    }
    const bf16x8* __restrict__ Ab = reinterpret_cast<const bf16x8*>(&Abuf[buf][half * R + l31]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[NT], am[NT], al[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        ah[t] = Ab[(0 * 4 + 2 * s) * R + t * 32];
        am[t] = Ab[(1 * 4 + 2 * s) * R + t * 32];
        al[t] = Ab[(2 * 4 + 2 * s) * R + t * 32];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], Bm[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], Bh[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], Bl[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], Bh[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], Bm[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], Bh[s], acc[t], 0, 0, 0);
    }
    if (kc + 1 < nch) {
      split8(raw[0], Bh[0], Bm[0], Bl[0]);
      split8(raw[1], Bh[1], Bm[1], Bl[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next weight chunk has landed
    __syncthreads();
  }

  // ---- epilogue: bias, residual or gate, store -------------------------------------------------------------
  float* __restrict__ yb = a.y.p + (long)b * a.y.bs;
  const __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
  int pv = pix < P ? (4 * half * P + pix) * 4 : OOB;
  if constexpr (EPI == EP_UNSHUFFLE) {
    // pixel-unshuffle(2): output channel 4 row + 2 (y & 1) + (x & 1) at (y >> 1, x >> 1); the row part of the address
    // is the same scalar as for a plain store (4 row planes of P / 4 = row * P)
    const int py = pix / a.W, px = pix - py * a.W;
    pv = pix < P ? ((2 * (py & 1) + (px & 1) + 16 * half) * (P >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1)) * 4 : OOB;
  }
  __amdgpu_buffer_rsrc_t r_rsrc = y_rsrc;
  if constexpr (EPI == EP_RESID || EPI == EP_GELU_MUL)
    r_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.r.p + (long)b * a.r.bs, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
  if constexpr (EPI == EP_AV_INTRA) {
    // y = acc; y2 = acc + q[src] * rq[src] + k[src] * rk[src] with src = the order's entry for the output row
    const int* __restrict__ perm = a.kidx + (long)(b / a.imgs_per_group) * M;
    const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.q.p + (long)b * a.q.bs, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.k.p + (long)b * a.k.bs, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t y2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.y2.p + (long)b * a.y2.bs, 0, (int)min((long)M * P * 4, 0x7fffffffL), 0x00020000);
    const float* __restrict__ rq = a.rq + (long)b * M;
    const float* __restrict__ rk = a.rk + (long)b * M;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int m0 = t * 32;
      if (m0 >= M) continue;
      float qv[16], kv[16], sq[16], sk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2);
        const int s0 = perm[min(row, M - 1)], s1 = perm[min(row + 4, M - 1)];
        const int src = half ? s1 : s0;
        const int vo = row + 4 * half < M && pix < P ? (src * P + pix) * 4 : OOB;
        qv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(q_rsrc, vo, 0, 0));
        kv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(k_rsrc, vo, 0, 0));
        sq[r] = half ? rq[s1] : rq[s0];
        sk[r] = half ? rk[s1] : rk[s0];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2);
        const int vo = row + 4 * half < M ? pv : OOB;
        const float v = acc[t][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, vo, min(row, M - 1) * P * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v + qv[r] * sq[r] + kv[r] * sk[r]), y2_rsrc, vo,
                                              min(row, M - 1) * P * 4, 0);
      }
    }
    return;
  }
  const float* __restrict__ bias = a.bias ? a.bias + (long)b * a.bias_bs : nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int m0 = t * 32;
    if (m0 >= M) continue;
    float aux[16], bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2);
      const int vo = row + 4 * half < M ? pv : OOB;    // rows past M: nothing read, nothing stored
      bv[r] = bias ? bias[min(row + 4 * half, M - 1)] : 0.f;
      if constexpr (EPI == EP_RESID || EPI == EP_GELU_MUL)
        aux[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vo, min(row, M - 1) * P * 4, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2);
      const int vo = row + 4 * half < M ? pv : OOB;
      float v = acc[t][r] + bv[r];
      if constexpr (EPI == EP_RESID) v += aux[r];
      if constexpr (EPI == EP_GELU_MUL) v = gelu_fast(v) * aux[r];
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_rsrc, vo, min(row, M - 1) * P * 4, 0);
    }
  }
}

static std::atomic<int>& split_flag() {
  static std::atomic<int> on{std::getenv("JIRF_GEMM_SPLIT_BF16") ? std::atoi(std::getenv("JIRF_GEMM_SPLIT_BF16")) != 0 : 1};
  return on;
}
bool gemm_split_bf16_enabled() { return split_flag().load(std::memory_order_relaxed) != 0; }
extern "C" int jirf_set_gemm_split_bf16(int enable) { return split_flag().exchange(enable != 0); }
bool gemm_split_bf16_covers(int K) { return K == 32 || K == 48 || K == 64 || K == 96; }

static uint16_t bf16_rne(float v) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float v;
  std::memcpy(&v, &u, 4);
  return v;
}

size_t pack_wt_split3(const float* w, int M, int K, int Mpad, std::vector<uint16_t>& out) {
  const int kp = (K + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
  const size_t plane = (size_t)kp * Mpad;
  out.assign(3 * plane, 0);
  for (int o = 0; o < M; ++o)
    for (int k = 0; k < K; ++k) {
      const float v = w[(size_t)o * K + k];
      const uint16_t h = bf16_rne(v);
      const float r1 = v - bf16_to_f32(h);
      const uint16_t m = bf16_rne(r1);
      const uint16_t l = bf16_rne(r1 - bf16_to_f32(m));
      const size_t at = ((size_t)(k / 8) * Mpad + o) * 8 + (k % 8);
      out[at] = h; out[plane + at] = m; out[2 * plane + at] = l;
    }
  return plane;
}

size_t pack_split3_kmajor(const std::vector<float>& packed, int Kp, int Mpad, std::vector<uint16_t>& out) {
  const size_t plane = (size_t)Kp * Mpad;
  out.assign(3 * plane, 0);
  for (int k = 0; k < Kp; ++k)
    for (int o = 0; o < Mpad; ++o) {
      const float v = packed[(size_t)k * Mpad + o];
      const uint16_t h = bf16_rne(v);
      const float r1 = v - bf16_to_f32(h);
      const uint16_t m = bf16_rne(r1);
      const uint16_t l = bf16_rne(r1 - bf16_to_f32(m));
      const size_t at = ((size_t)(k / 8) * Mpad + o) * 8 + (k % 8);
      out[at] = h; out[plane + at] = m; out[2 * plane + at] = l;
    }
  return plane;
}

// LayerNorm or plain 1x1 GEMM with a plain store (plain: also GELU, and a second source for the first rows); false when the shape is not one this kernel is built for.
bool gemm_b3_covers(const GemmArgs& a, int loader) {
  if (!gemm_split_bf16_enabled() || !a.wt3 || a.wt_bs != 0) return false;
  if (loader != LD_LN && loader != LD_PLAIN) return false;
  if (a.epi != EP_STORE && !(loader == LD_PLAIN && a.epi == EP_GELU)) return false;
  if (a.k_split > 0) {   // a second source: same arena, behind the first, within a descriptor's reach
    if (loader != LD_PLAIN || a.k_split % 8 != 0 || a.x.bs != a.x2.bs || a.x2.p < a.x.p) return false;
    if ((long)(a.x2.p - a.x.p) * 4 + (long)a.K * a.P * 4 > 0x7fffffffL) return false;
  }
  if (!gemm_split_bf16_covers(a.K) || a.M > 4096) return false;   // (the output bias of all rows waits in LDS)
  if (loader == LD_LN && a.lnw) return false;   // the affine part must be folded into the weights
  return (long)a.M * a.P * 4 <= 0x7fffffffL && (long)a.K * a.P * 4 <= 0x7fffffffL;
}

template <int W>
static void launch_gemm_b3_w(const GemmArgs& a, int loader, hipStream_t s) {
  const int tiles_n = (a.P + 32 * W - 1) / (32 * W);
  const size_t smem = (size_t)((a.M + 31) / 32 * 32) * sizeof(float);
  const dim3 grid((unsigned)(tiles_n * a.B)), block(64 * W);
#define JIRF_B3_CASE(NS)                                                                             \
  case NS:                                                                                           \
    if (loader == LD_LN) hipLaunchKernelGGL((gemm_b3_kernel<NS, true, W, false>), grid, block, smem, s, a, tiles_n);   \
    else if (a.epi == EP_GELU) hipLaunchKernelGGL((gemm_b3_kernel<NS, false, W, true>), grid, block, smem, s, a, tiles_n); \
    else hipLaunchKernelGGL((gemm_b3_kernel<NS, false, W, false>), grid, block, smem, s, a, tiles_n);    \
    break
  switch (a.K / 16) {
    JIRF_B3_CASE(2);
    JIRF_B3_CASE(3);
    JIRF_B3_CASE(4);
    JIRF_B3_CASE(6);
    default: JIRF_REQUIRE(false, JIRF_EINVAL, "launch_gemm_b3: K=%d", a.K);
  }
#undef JIRF_B3_CASE
  JIRF_HIP_CHECK(hipGetLastError());
}
void launch_gemm_b3(const GemmArgs& a, int loader, hipStream_t s) {
  if (b3_waves(a.P, a.B) == 8) launch_gemm_b3_w<8>(a, loader, s);
  else launch_gemm_b3_w<4>(a, loader, s);
}

// per-image fp32 matrices -> bf16 planes: thread (image, k group, column) reads its eight k and writes three 16-byte words
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w, u32x4* __restrict__ planes,
                                                           const int B, const int Kp, const int Mpad) {
  const long n = (long)B * (Kp / 8) * Mpad;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int m = (int)(e % Mpad);
  const long bg = e / Mpad;
  const int g = (int)(bg % (Kp / 8)), b = (int)(bg / (Kp / 8));
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = w[((long)b * Kp + 8 * g + i) * Mpad + m];
  bf16x8 h, mi, l;
  split8(v, h, mi, l);
  const long plane16 = (long)(Kp / 8) * Mpad;
  u32x4* __restrict__ dst = planes + (long)b * 3 * plane16 + (long)g * Mpad + m;
  dst[0] = __builtin_bit_cast(u32x4, h);
  dst[plane16] = __builtin_bit_cast(u32x4, mi);
  dst[2 * plane16] = __builtin_bit_cast(u32x4, l);
}
void launch_split_planes(const float* w, uint16_t* planes, int B, int Kp, int Mpad, hipStream_t s) {
  const long n = (long)B * (Kp / 8) * Mpad;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w,
                     reinterpret_cast<u32x4*>(planes), B, Kp, Mpad);
  JIRF_HIP_CHECK(hipGetLastError());
}

// short outputs (M <= 96): 1x1 with plain store / residual / GELU gate, or a dense k x k convolution (Cin % 16 == 0,
// more outputs than the few-output kernel takes) with plain store / residual / pixel-unshuffle
bool gemm_k3_covers(const GemmArgs& a, int loader) {
  if (!gemm_split_bf16_enabled() || !a.wt3 || a.k_split > 0 || a.M > 96) return false;
  if ((a.wt_bs != 0) != (a.wt3_bs != 0)) return false;                 // per-image weights need per-image planes
  if (a.wt_bs != 0 && loader != LD_GATHER && !(loader == LD_PLAIN && a.epi == EP_STORE)) return false;
  if ((long)((a.P + 127) / 128) * a.B < 512) return false;             // a grid that fills the chip
  if ((long)a.M * a.P * 4 > 0x7fffffffL) return false;
  if (loader == LD_CONV3) {
    if (a.epi != EP_STORE && a.epi != EP_RESID && a.epi != EP_UNSHUFFLE) return false;
    return a.M >= 17 && a.Cin % 16 == 0 && (long)a.Cin * a.Hin * a.Win * 4 <= 0x7fffffffL;
  }
  if (loader == LD_GATHER)   // attention * V with its fused tail: square, a channel order for rows and outputs
    return a.epi == EP_AV_INTRA && a.kidx && a.K == a.M && a.K >= 32 && !a.bias && (long)a.K * a.P * 4 <= 0x7fffffffL;
  if (loader != LD_PLAIN) return false;
  if (a.epi != EP_STORE && a.epi != EP_RESID && a.epi != EP_GELU_MUL) return false;
  return a.K >= (a.wt_bs != 0 ? 32 : 96) && (long)a.K * a.P * 4 <= 0x7fffffffL;
}

template <int W>
static void launch_gemm_k3_w(const GemmArgs& a, int loader, hipStream_t s) {
  const int tiles_n = (a.P + 32 * W - 1) / (32 * W);
  const dim3 grid((unsigned)(tiles_n * a.B)), block(64 * W);
  const int NT = (a.M + 31) / 32;
#define JIRF_K3_EPI(N)                                                                                              \
  do {                                                                                                              \
    if (loader == LD_GATHER) {                                                                                      \
      hipLaunchKernelGGL((gemm_k3_kernel<N, EP_AV_INTRA, W, LD_GATHER>), grid, block, 0, s, a, tiles_n);            \
    } else if (loader == LD_CONV3) {                                                                                \
      if (a.epi == EP_RESID) hipLaunchKernelGGL((gemm_k3_kernel<N, EP_RESID, W, LD_CONV3>), grid, block, 0, s, a, tiles_n);   \
      else if (a.epi == EP_UNSHUFFLE) hipLaunchKernelGGL((gemm_k3_kernel<N, EP_UNSHUFFLE, W, LD_CONV3>), grid, block, 0, s, a, tiles_n); \
      else hipLaunchKernelGGL((gemm_k3_kernel<N, EP_STORE, W, LD_CONV3>), grid, block, 0, s, a, tiles_n);            \
    } else if (a.epi == EP_RESID) hipLaunchKernelGGL((gemm_k3_kernel<N, EP_RESID, W, LD_PLAIN>), grid, block, 0, s, a, tiles_n); \
    else if (a.epi == EP_GELU_MUL) hipLaunchKernelGGL((gemm_k3_kernel<N, EP_GELU_MUL, W, LD_PLAIN>), grid, block, 0, s, a, tiles_n); \
    else hipLaunchKernelGGL((gemm_k3_kernel<N, EP_STORE, W, LD_PLAIN>), grid, block, 0, s, a, tiles_n);              \
  } while (0)
  if (NT == 1) JIRF_K3_EPI(1);
  else if (NT == 2) JIRF_K3_EPI(2);
  else JIRF_K3_EPI(3);
#undef JIRF_K3_EPI
  JIRF_HIP_CHECK(hipGetLastError());
}
void launch_gemm_k3(const GemmArgs& a, int loader, hipStream_t s) {
  if (b3_waves(a.P, a.B) == 8) launch_gemm_k3_w<8>(a, loader, s);
  else launch_gemm_k3_w<4>(a, loader, s);
}

}  // namespace jirf
