// The gated feed-forward block of the Restormer family as ONE kernel (gfx950):
//
//     y = x + W_out ( gelu(dw3x3(T)[:h]) * dw3x3(T)[h:] ),      T = W_in LayerNorm(x)
//
// (HINT FeedForward.execute hint_arch.py:150-186 behind norm2 and the residual of TransformerBlock :408-434).  As
// three launches the hidden tensor T (2h = 510 channels where x has 96) makes three trips through HBM; here it
// never leaves the compute unit.
//
// Line-buffer form.  A workgroup of eight waves owns a vertical strip of the image, 32 pixel columns wide (30
// output columns + one halo column each side, so the 3x3 taps find their neighbours inside the strip), and rolls
// down it one image row per step.  Wave w owns hidden channels 32 w .. 32 w + 31 of both halves of T for the
// whole strip, and per step (waves 4-7 in another order, see the kernel)
//   1. forms row y + 1 of its 64 channels of T on the matrix cores (v_mfma_f32_32x32x2_f32: A = its rows of W_in
//      from L2 in a lane-major packing, B = the normalised input row, shared by the eight waves through LDS);
//      pixel columns sit on the MFMA column (lane) index, channels in the 16 accumulator registers;
//   2. convolves row y from its ring of three accumulator rows: vertical taps are other registers of the same
//      lane, horizontal taps whole-wave DPP shifts, weights broadcast reads from LDS; then the gate;
//   3. multiplies its gated 32 x 32 chunk by its 32 columns of W_out.  The chunk is used where it lies: register
//      r of an accumulator tile holds hidden row 8 (r >> 2) + (r & 3) + 4 (lane >> 5), which is exactly the
//      B-operand lane map of a k-step over that pair of rows, so the sixteen registers are sixteen k-steps and
//      nothing is transposed or staged.  The eight waves' partial 96 x 32 output tiles are added through LDS in
//      wave order (deterministic), the residual is added and the row stored.
// Every wave also loads and normalises four pixels of the next input row (statistics in registers, lane exchange
// on the vector ALU) into the LDS slot the matrix phase reads one step later.
// Halo cost: 32 columns computed for 30 stored, two extra rows per segment of a strip.  Pixels outside the image
// must see T = 0 (the depthwise convolution pads T, not x): rows outside are skipped, columns outside are
// multiplied by a per-lane mask.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "runtime.h"

namespace jirf {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int GF_C = 96;                 // channels of x
constexpr int GF_G = GF_C / 8;           // groups of four k-steps (a k-step = 2 channels)
constexpr int GF_NW = 8;                 // waves = hidden chunks of 32
constexpr int GF_OW = 30;                // output columns per strip
constexpr int GF_XH = GF_G * 2 * 32 * 4; // floats of one normalised row in LDS: [group][half][pixel][4 k-steps]
constexpr int GF_DW = GF_NW * 2 * 9 * 4 * 2 * 4;   // depthwise weights [wave][part][tap][q][half][4]
constexpr int GF_BI = GF_NW * 2 * 4 * 2 * 4;       // a bias table [wave][part][q][half][4]
constexpr int GF_PART = GF_NW * 3 * 4 * 64 * 4;    // partial output tiles [wave][m-tile][q][lane][4]
constexpr int GF_LDS_FLOATS = 2 * GF_XH + GF_DW + 2 * GF_BI + GF_PART;

struct GdfnArgs {
  const float* x; long x_bs;
  float* y; long y_bs;
  int B, H, W;
  int nsx, seg, nseg;          // strips per image row, rows per segment, segments per strip
  int ln_center;               // WithBias LayerNorm subtracts the mean, BiasFree does not
  const float *win, *bin, *dwp, *dwb, *wout, *bout;
};

__device__ __forceinline__ float dpp_from_left(float v) {    // lane i takes lane i - 1 (wave_shr:1)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {   // lane i takes lane i + 1 (wave_shl:1)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__global__ __launch_bounds__(512) void gdfn_fused_kernel(const GdfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gf_smem[];
  float* __restrict__ Xh = gf_smem;                   // [2][GF_XH]
  float* __restrict__ dw_s = Xh + 2 * GF_XH;
  float* __restrict__ bin_s = dw_s + GF_DW;
  float* __restrict__ dwb_s = bin_s + GF_BI;
  float* __restrict__ part = dwb_s + GF_BI;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W;
  const long P = (long)H * W;

  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int per_img = a.nsx * a.nseg;
  const int b = id / per_img;
  id -= b * per_img;
  const int sgm = id / a.nsx, strip = id - sgm * a.nsx;
  const int c_in0 = strip * GF_OW - 1;                  // image column of the strip's first (halo) column
  const int ry0 = sgm * a.seg, ry1 = min(H, ry0 + a.seg);
  const float* __restrict__ xb = a.x + (long)b * a.x_bs;
  float* __restrict__ yb = a.y + (long)b * a.y_bs;

  for (int i = tid; i < GF_DW; i += 512) dw_s[i] = a.dwp[i];
  if (tid < GF_BI) {
    bin_s[tid] = a.bin[tid];
    dwb_s[tid] = a.dwb ? a.dwb[tid] : 0.f;
  }
  const int pxl = c_in0 + l31;
  const float colmask = (pxl >= 0 && pxl < W) ? 1.f : 0.f;
  const bool edge = c_in0 < 0 || c_in0 + 32 > W;        // the strip has columns outside the image

  // ---- LayerNorm of one input row into an LDS slot (all waves) ---------------------------------------------
  // Wave w normalises pixels 4 w .. 4 w + 3 of the row: lane = channel lane (16) x pixel (4), a lane holds channels
  // cl + 16 u, u < 6, of one pixel as six dwords.  Mean and variance (two passes, as the reference's mean /
  // var(unbiased=False)) are sums over the lane bits 2..5 on the vector ALU.  Six registers from request to use,
  // the same small share for every wave, no alignment demands on the row, and a pixel outside the image is a zero.
  const int ln_cl = lane >> 2, ln_px = 4 * wave + (lane & 3);
  const int ln_col = c_in0 + ln_px;
  const bool ln_ok = ln_col >= 0 && ln_col < W;
  const float* __restrict__ ln_src = xb + (long)ln_cl * P + (ln_ok ? ln_col : 0);
  auto ln_issue = [&](int row, float (&v)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = ln_src[(long)(16 * u) * P + (long)row * W];
  };
  auto sum_bits2345 = [](float t) __attribute__((always_inline)) {
    auto bits = [](float f) { return __builtin_bit_cast(unsigned, f); };
    auto flt = [](unsigned u) { return __builtin_bit_cast(float, u); };
    t += flt(__builtin_amdgcn_update_dpp(0, bits(t), 0x124, 0xf, 0xf, false));   // row_ror:4
    t += flt(__builtin_amdgcn_update_dpp(0, bits(t), 0x128, 0xf, 0xf, false));   // row_ror:8
    const auto s16 = __builtin_amdgcn_permlane16_swap(bits(t), bits(t), false, false);
    t = flt(s16[0]) + flt(s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(bits(t), bits(t), false, false);
    return flt(s32[0]) + flt(s32[1]);
  };
  auto ln_finish = [&](float (&v)[6], int slot) __attribute__((always_inline)) {
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      if (!ln_ok) v[u] = 0.f;
      sum += v[u];
    }
    const float mu = sum_bits2345(sum) * (1.0f / GF_C);
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) q = fmaf(v[u] - mu, v[u] - mu, q);
    const float rs = __builtin_amdgcn_rsqf(sum_bits2345(q) * (1.0f / GF_C) + 1e-5f);
    const float mc = a.ln_center ? mu : 0.f;
    // channel k = cl + 16 u is k-step k >> 1 (group k >> 3, position (k >> 1) & 3) of lane half k & 1
    float* __restrict__ dst = Xh + slot * GF_XH + (((ln_cl >> 3) * 2 + (ln_cl & 1)) * 32 + ln_px) * 4 + ((ln_cl >> 1) & 3);
#pragma unroll
    for (int u = 0; u < 6; ++u) dst[u * 2 * 2 * 32 * 4] = (v[u] - mc) * rs;
  };

  // ---- ring of three rows of this wave's 64 hidden channels (accumulator layout) ---------------------------
  f32x16 T[3][2];
  const f32x4* __restrict__ win_w0 = reinterpret_cast<const f32x4*>(a.win) + (long)wave * 2 * GF_G * 64 + lane;
  const f32x4* __restrict__ wout_w0 = reinterpret_cast<const f32x4*>(a.wout) + (long)wave * 3 * 4 * 64 + lane;
  const f32x4* __restrict__ bin_w0 = reinterpret_cast<const f32x4*>(bin_s) + wave * 2 * 4 * 2 + half;
  const f32x4* __restrict__ dwb_w0 = reinterpret_cast<const f32x4*>(dwb_s) + wave * 2 * 4 * 2 + half;
  const f32x4* __restrict__ dw_w0 = reinterpret_cast<const f32x4*>(dw_s) + wave * 2 * 9 * 4 * 2 + half;

  // reduce phase: this thread owns two adjacent registers (rows) of one lane position, for each of the 3 m-tiles
  const int rd_i2 = tid & 1, rd_lane = (tid >> 1) & 63, rd_q = (tid >> 7) & 3;
  const int rd_l31 = rd_lane & 31, rd_half = rd_lane >> 5;
  const int rd_col = c_in0 + rd_l31;
  const bool rd_ok = rd_l31 >= 1 && rd_l31 <= GF_OW && rd_col < W;
  const int rd_row = 8 * rd_q + 2 * rd_i2 + 4 * rd_half;      // first of the two output channels inside an m-tile
  const f32x2* __restrict__ rd_src = reinterpret_cast<const f32x2*>(part) + (rd_q * 64 + rd_lane) * 2 + rd_i2;

  // first input row of the segment
  if (ry0 - 1 >= 0) {
    float v[6];
    ln_issue(ry0 - 1, v);
    ln_finish(v, (ry0 - 1) & 1);
  }
  __syncthreads();

  // The two waves that share a SIMD (w and w + 4) must not do the same kind of work at the same time: the three
  // phases of a step - row of T (matrix), convolution + gate (vector), output product (matrix) - depend on each other
  // inside a wave, and two waves in the same phase leave one of the two pipes idle.  So the halves of the workgroup
  // run the same step in a different order, one output row apart:
  //   waves 0-3   row yi of T,  then convolution + product of output row yi - 1
  //   waves 4-7   convolution + product of output row yi - 2,  then row yi of T
  // (either order needs three ring rows).  Their partial tiles of one output row therefore arrive a step apart: the
  // reducing threads keep the sum over waves 0-3 in registers for one step and finish the row when waves 4-7
  // deliver theirs.  The sum is taken in a fixed order either way.
  const bool late = wave >= 4;
  f32x2 carry[3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) carry[mt] = f32x2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) { T[k][0][r] = 0.f; T[k][1][r] = 0.f; }
  auto step = [&](auto slot_tag, auto out_tag, int yi) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_tag)::value;          // ring slot of row yi; yi - 1 in (S + 2) % 3, yi - 2 in (S + 1) % 3
    constexpr int SM = (S + 2) % 3, SO = (S + 1) % 3;

    const bool row_in = yi >= 0 && yi < H && yi <= ry1;
    const int yo = yi - 2;                                // the row finished in this step
    const bool fin = yo >= ry0, out_early = yi - 1 >= ry0 && yi - 1 < ry1;
    // Every weight word below is the same in every step, so the compiler would read all of them once, ahead of the
    // row loop, into some four hundred registers, and spill the ring to make room.  An offset it cannot see through
    // (always 0) ties each read to its own step.
    int z = 0;
    asm volatile("" : "+s"(z));
    const f32x4* __restrict__ win_w = win_w0 + z;
    const f32x4* __restrict__ wout_w = wout_w0 + z;
    const f32x4* __restrict__ bin_w = bin_w0 + z;
    const f32x4* __restrict__ dwb_w = dwb_w0 + z;
    const f32x4* __restrict__ dw_w = dw_w0 + z;
    const int yn = yi + 1;
    const bool next_in = yn <= ry1 && yn < H;           // the next row exists: normalise it during this step

    // the next input row is requested here, behind the last weight words of the matrix phase and in front of the
    // convolution, which reads LDS only: the vector-memory counter is in order, so whatever is requested behind these
    // loads waits for them too
    float lnv[6];
    float res[3][2];
    if (next_in) ln_issue(yn, lnv);

    if (!late) {
    // ---- 1. row yi of T ----
    if (row_in) {
      f32x16 a1, a2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b1 = bin_w[(0 * 4 + q) * 2], b2 = bin_w[(1 * 4 + q) * 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a1[4 * q + i] = b1[i]; a2[4 * q + i] = b2[i]; }
      }
      const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(Xh + (yi & 1) * GF_XH) + half * 32 + l31;
      // weight words two groups ahead and the normalised row one group ahead, in program order (see the output
      // product below for why)
      f32x4 wa0 = win_w[0], wb0 = win_w[GF_G * 64], wa1 = win_w[64], wb1 = win_w[(GF_G + 1) * 64];
      f32x4 bq0 = bp[0];
#pragma unroll
      for (int g = 0; g < GF_G; ++g) {
        f32x4 wa2 = wa1, wb2 = wb1, bq1 = bq0;
        if (g + 2 < GF_G) { wa2 = win_w[(g + 2) * 64]; wb2 = win_w[(GF_G + g + 2) * 64]; }
        if (g + 1 < GF_G) bq1 = bp[(g + 1) * 64];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0[j], bq0[j], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb0[j], bq0[j], a2, 0, 0, 0);
        }
        wa0 = wa1; wb0 = wb1; wa1 = wa2; wb1 = wb2; bq0 = bq1;
      }
      if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { a1[r] *= colmask; a2[r] *= colmask; }
      }
      T[S][0] = a1;
      T[S][1] = a2;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { T[S][0][r] = 0.f; T[S][1][r] = 0.f; }
    }

    if (out_early) {
      // ---- 2. depthwise 3x3 of row yo (rows yo - 1, yo, yo + 1 = slots SO, SM, S) and the gate ----
      f32x16 G;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // a fence for the compiler only: left alone the scheduler requests all 72 weight words of the row up
          // front and the register allocator evicts the ring to scratch to hold them
          asm volatile("" ::: "memory");
          d[t] = dwb_w[(t * 4 + q) * 2];
          f32x4 wrow[3], wnext[3];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) wrow[kx] = dw_w[((t * 9 + kx) * 4 + q) * 2];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            if (ky < 2) {
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) wnext[kx] = dw_w[((t * 9 + (ky + 1) * 3 + kx) * 4 + q) * 2];
            }
            asm volatile("" ::: "memory");
            const f32x16& src = ky == 0 ? T[SO][t] : ky == 1 ? T[SM][t] : T[S][t];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const f32x4 wv = wrow[kx];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float c = src[4 * q + i];
                const float sh = kx == 0 ? dpp_from_left(c) : kx == 2 ? dpp_from_right(c) : c;
                d[t][i] = fmaf(wv[i], sh, d[t][i]);
              }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wrow[kx] = wnext[kx];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) G[4 * q + i] = gelu_fast(d[0][i]) * d[1][i];

      }
      // the residual of the row finished in this step (row 0 when there is none yet: read and not used - a branch
      // here would cut the step into blocks, across which the register allocator does much worse): requested behind
      // the convolution's registers, in front of the 48 MFMAs
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          res[mt][e] = xb[(long)(32 * mt + rd_row + e) * P + (long)max(yo, 0) * W + (rd_ok ? rd_col : 0)];
      // ---- 3. this wave's share of W_out G: 16 k-steps per 32 output channels ----
      // (the weight words two groups ahead, in program order: where the registers are this tight the compiler keeps
      // the order it is given, and a load placed next to its use waits a whole L2 round trip in front of every four
      // matrix instructions)
      f32x4 wo0 = wout_w[0], wo1 = wout_w[64];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 wo2 = wo1;
          if (mt * 4 + q + 2 < 12) wo2 = wout_w[(mt * 4 + q + 2) * 64];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(wo0[i], G[4 * q + i], o, 0, 0, 0);
          wo0 = wo1;
          wo1 = wo2;
        }
        f32x4* __restrict__ pd = reinterpret_cast<f32x4*>(part) + ((wave * 3 + mt) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) pd[q * 64] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
      }
    }

    if (!out_early && fin) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          res[mt][e] = xb[(long)(32 * mt + rd_row + e) * P + (long)yo * W + (rd_ok ? rd_col : 0)];
    }
    } else {
    if (fin) {
      // ---- 2. depthwise 3x3 of row yo (rows yo - 1, yo, yo + 1 = slots SO, SM, S) and the gate ----
      f32x16 G;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // a fence for the compiler only: left alone the scheduler requests all 72 weight words of the row up
          // front and the register allocator evicts the ring to scratch to hold them
          asm volatile("" ::: "memory");
          d[t] = dwb_w[(t * 4 + q) * 2];
          f32x4 wrow[3], wnext[3];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) wrow[kx] = dw_w[((t * 9 + kx) * 4 + q) * 2];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            if (ky < 2) {
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) wnext[kx] = dw_w[((t * 9 + (ky + 1) * 3 + kx) * 4 + q) * 2];
            }
            asm volatile("" ::: "memory");
            const f32x16& src = ky == 0 ? T[S][t] : ky == 1 ? T[SO][t] : T[SM][t];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const f32x4 wv = wrow[kx];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float c = src[4 * q + i];
                const float sh = kx == 0 ? dpp_from_left(c) : kx == 2 ? dpp_from_right(c) : c;
                d[t][i] = fmaf(wv[i], sh, d[t][i]);
              }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wrow[kx] = wnext[kx];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) G[4 * q + i] = gelu_fast(d[0][i]) * d[1][i];

      }
      // the residual of the row finished in this step (row 0 when there is none yet: read and not used - a branch
      // here would cut the step into blocks, across which the register allocator does much worse): requested behind
      // the convolution's registers, in front of the 48 MFMAs
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          res[mt][e] = xb[(long)(32 * mt + rd_row + e) * P + (long)max(yo, 0) * W + (rd_ok ? rd_col : 0)];
      // ---- 3. this wave's share of W_out G: 16 k-steps per 32 output channels ----
      // (the weight words two groups ahead, in program order: where the registers are this tight the compiler keeps
      // the order it is given, and a load placed next to its use waits a whole L2 round trip in front of every four
      // matrix instructions)
      f32x4 wo0 = wout_w[0], wo1 = wout_w[64];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 wo2 = wo1;
          if (mt * 4 + q + 2 < 12) wo2 = wout_w[(mt * 4 + q + 2) * 64];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_32x32x2f32(wo0[i], G[4 * q + i], o, 0, 0, 0);
          wo0 = wo1;
          wo1 = wo2;
        }
        f32x4* __restrict__ pd = reinterpret_cast<f32x4*>(part) + ((wave * 3 + mt) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) pd[q * 64] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
      }
    }

    // ---- 1. row yi of T ----
    if (row_in) {
      f32x16 a1, a2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b1 = bin_w[(0 * 4 + q) * 2], b2 = bin_w[(1 * 4 + q) * 2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a1[4 * q + i] = b1[i]; a2[4 * q + i] = b2[i]; }
      }
      const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(Xh + (yi & 1) * GF_XH) + half * 32 + l31;
      // weight words two groups ahead and the normalised row one group ahead, in program order (see the output
      // product below for why)
      f32x4 wa0 = win_w[0], wb0 = win_w[GF_G * 64], wa1 = win_w[64], wb1 = win_w[(GF_G + 1) * 64];
      f32x4 bq0 = bp[0];
#pragma unroll
      for (int g = 0; g < GF_G; ++g) {
        f32x4 wa2 = wa1, wb2 = wb1, bq1 = bq0;
        if (g + 2 < GF_G) { wa2 = win_w[(g + 2) * 64]; wb2 = win_w[(GF_G + g + 2) * 64]; }
        if (g + 1 < GF_G) bq1 = bp[(g + 1) * 64];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0[j], bq0[j], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb0[j], bq0[j], a2, 0, 0, 0);
        }
        wa0 = wa1; wb0 = wb1; wa1 = wa2; wb1 = wb2; bq0 = bq1;
      }
      if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { a1[r] *= colmask; a2[r] *= colmask; }
      }
      T[S][0] = a1;
      T[S][1] = a2;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { T[S][0][r] = 0.f; T[S][1][r] = 0.f; }
    }

    }
    if (next_in) ln_finish(lnv, yn & 1);
    __syncthreads();

    // the partial tiles: waves 0-3 delivered row yi - 1 (kept for one step), waves 4-7 row yi - 2 (finished now)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      f32x2 early = rd_src[(0 * 3 + mt) * 4 * 64 * 2];
#pragma unroll
      for (int w = 1; w < 4; ++w) early += rd_src[(w * 3 + mt) * 4 * 64 * 2];
      if (fin) {
        f32x2 sl = rd_src[(4 * 3 + mt) * 4 * 64 * 2];
#pragma unroll
        for (int w = 5; w < GF_NW; ++w) sl += rd_src[(w * 3 + mt) * 4 * 64 * 2];
        sl += carry[mt];
        if (rd_ok) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int co = 32 * mt + rd_row + e;
            float v = sl[e] + res[mt][e];
            if (a.bout) v += a.bout[co];
            yb[(long)co * P + (long)yo * W + rd_col] = v;
          }
        }
      }
      carry[mt] = early;
    }
    __syncthreads();
  };

  // One loop body: the ring turns by register moves (96 per step) instead of by three unrolled copies of the step with
  // rotated slot numbers - a third of the code.
  for (int yi = ry0 - 1; yi <= ry1 + 1; ++yi) {
    step(std::integral_constant<int, 0>{}, std::false_type{}, yi);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x16 oldest = T[1][t];   // row yi - 2: waves 4-7 still read it (from slot 0) in the next step
      T[1][t] = T[2][t];
      T[2][t] = T[0][t];
      T[0][t] = oldest;
    }
  }
}

DynLdsOnce g_gdfn_lds;


}  // namespace

// ---- host side: packing and launch ---------------------------------------------------------------------------
bool gdfn_fused_supported(int C, int hidden, int H, int W) {
  return C == GF_C && hidden >= 1 && hidden <= 32 * GF_NW && W >= 32 && H >= 1 && (long)H * W * GF_C < (1L << 40);
}

GdfnFusedW pack_upload_gdfn_fused(WeightPool& pool, int C, int hidden, const float* ln_w, const float* ln_b,
                                  const float* w_in, const float* b_in, const float* w_dw, const float* b_dw,
                                  const float* w_out, const float* b_out) {
  JIRF_REQUIRE(C == GF_C && hidden >= 1 && hidden <= 32 * GF_NW, JIRF_EINVAL, "fused GDFN: C=%d hidden=%d is not built",
               C, hidden);
  const int h = hidden;
  // W_in with the LayerNorm weight folded in; its bias = W_in ln_b (+ b_in), summed in double
  auto win = [&](int o, int k) { return w_in[(long)o * C + k] * (ln_w ? ln_w[k] : 1.f); };
  std::vector<float> win_pk((size_t)GF_NW * 2 * GF_G * 64 * 4, 0.f), bin_pk(GF_BI, 0.f), dw_pk(GF_DW, 0.f),
      dwb_pk(GF_BI, 0.f), wout_pk((size_t)GF_NW * 3 * 4 * 64 * 4, 0.f);
  for (int w = 0; w < GF_NW; ++w)
    for (int t = 0; t < 2; ++t) {
      for (int g = 0; g < GF_G; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int ch = 32 * w + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
            if (ch < h) win_pk[((((size_t)w * 2 + t) * GF_G + g) * 64 + lane) * 4 + j] = win(t * h + ch, k);
          }
      for (int q = 0; q < 4; ++q)
        for (int hf = 0; hf < 2; ++hf)
          for (int i = 0; i < 4; ++i) {
            const int ch = 32 * w + 8 * q + 4 * hf + i;
            if (ch >= h) continue;
            const int o = t * h + ch;
            double bsum = b_in ? (double)b_in[o] : 0.0;
            if (ln_b)
              for (int k = 0; k < C; ++k) bsum += (double)w_in[(long)o * C + k] * (double)ln_b[k];
            const size_t at = ((((size_t)w * 2 + t) * 4 + q) * 2 + hf) * 4 + i;
            bin_pk[at] = (float)bsum;
            dwb_pk[at] = b_dw ? b_dw[o] : 0.f;
            for (int tap = 0; tap < 9; ++tap)
              dw_pk[(((((size_t)w * 2 + t) * 9 + tap) * 4 + q) * 2 + hf) * 4 + i] = w_dw[(long)o * 9 + tap];
          }
    }
  for (int w = 0; w < GF_NW; ++w)
    for (int mt = 0; mt < 3; ++mt)
      for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 4; ++i) {
            const int co = 32 * mt + (lane & 31), ch = 32 * w + 8 * q + i + 4 * (lane >> 5);
            if (ch < h) wout_pk[((((size_t)w * 3 + mt) * 4 + q) * 64 + lane) * 4 + i] = w_out[(long)co * h + ch];
          }
  GdfnFusedW r;
  r.C = C;
  r.hidden = h;
  r.win = pool.upload(win_pk);
  r.bin = pool.upload(bin_pk);
  r.dwp = pool.upload(dw_pk);
  r.dwb = b_dw ? pool.upload(dwb_pk) : nullptr;
  r.wout = pool.upload(wout_pk);
  r.bout = b_out ? pool.upload(b_out, (size_t)C) : nullptr;
  return r;
}


void launch_gdfn_fused(const GdfnFusedW& w, TV x, TV y, int B, int H, int W, bool ln_center, hipStream_t s) {
  JIRF_REQUIRE(gdfn_fused_supported(w.C, w.hidden, H, W), JIRF_EINVAL, "fused GDFN: %dx%d with C=%d hidden=%d is not built",
               H, W, w.C, w.hidden);
  JIRF_REQUIRE(x.p != y.p, JIRF_EINVAL, "fused GDFN: input and output must be different buffers (strips read their neighbours' columns)");
  GdfnArgs a{};
  a.x = x.p; a.x_bs = x.bs; a.y = y.p; a.y_bs = y.bs;
  a.B = B; a.H = H; a.W = W;
  a.nsx = (W + GF_OW - 1) / GF_OW;
  // rows per segment: enough workgroups for a few rounds of the chip, as few segments as that allows (each
  // segment recomputes two halo rows of T)
  static const int seg_env = std::getenv("JIRF_GDFN_SEG") ? std::atoi(std::getenv("JIRF_GDFN_SEG")) : 0;
  int seg = H;
  if (seg_env > 0) {
    seg = std::min(H, seg_env);
  } else {
    while (seg > 16 && (long)B * a.nsx * ((H + seg - 1) / seg) < 4 * 256) seg = (seg + 1) / 2;
  }
  a.seg = seg;
  a.nseg = (H + seg - 1) / seg;
  a.ln_center = ln_center ? 1 : 0;
  a.win = w.win; a.bin = w.bin; a.dwp = w.dwp; a.dwb = w.dwb; a.wout = w.wout; a.bout = w.bout;
  const long P = (long)H * W;
  const int h = w.hidden;
  Scope sc(intern_label("gdfn_fused C%d h%d H%d W%d", w.C, h, H, W),
           (2.0 * w.C * 2 * h + 18.0 * 2 * h + 2.0 * h * w.C) * (double)P * B, 4.0 * B * (double)P * 2 * w.C, s);
  const int lds = GF_LDS_FLOATS * (int)sizeof(float);
  allow_dynamic_lds(g_gdfn_lds, reinterpret_cast<const void*>(gdfn_fused_kernel), lds);
  hipLaunchKernelGGL(gdfn_fused_kernel, dim3((unsigned)((long)B * a.nsx * a.nseg)), dim3(512), lds, s, a);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace jirf
