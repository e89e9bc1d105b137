// Script-level image plumbing of the reference's test scripts as gfx950 kernels, so a whole
// "load -> pad -> (tile) -> restore -> (merge) -> crop -> clamp -> uint8" pass stays on the device:
//   reflect pad to a multiple of 8      Deraining/test_rain100L.py:147-152
//   splitimage / mergeimage             Deraining/test_rain100L.py:31-75 (uniform score map: mean blend)
//   crop + clamp + img_as_ubyte         Deraining/test_rain100L.py:171-190
// All of it is index or byte work: results are bit-exact against the CPU restatement.
#include "common.h"

namespace jirf {

// y[b,c,yy,xx] = x[b,c,ry,rx] with reflection (no edge repeat) past the bottom / right border
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int H, int W, int Ho, int Wo, long planes) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = planes * Ho * Wo;
  if (i >= n) return;
  const int xx = (int)(i % Wo);
  const long t = i / Wo;
  const int yy = (int)(t % Ho);
  const long pl = t / Ho;
  const int ry = yy < H ? yy : 2 * (H - 1) - yy;
  const int rx = xx < W ? xx : 2 * (W - 1) - xx;
  y[i] = x[(pl * H + ry) * W + rx];
}

// tiles[(t*B + b), c, :, :] = x[b, c, hs[t/nw] : +crop, ws[t%nw] : +crop]   (tile-major like the
// reference's list of [B,C,crop,crop] crops, row starts outer, column starts inner)
__global__ __launch_bounds__(256) void split_tiles_kernel(const float* __restrict__ x, float* __restrict__ tiles,
                                                          TileGrid g, int B, int C, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_tile = (long)B * C * g.crop * g.crop;
  const long n = per_tile * g.nh * g.nw;
  if (i >= n) return;
  const int xx = (int)(i % g.crop);
  long t = i / g.crop;
  const int yy = (int)(t % g.crop);
  t /= g.crop;
  const int c = (int)(t % C);
  t /= C;
  const int b = (int)(t % B);
  const int tile = (int)(t / B);
  const int hs = g.hs[tile / g.nw], ws = g.ws[tile % g.nw];
  tiles[i] = x[(((long)b * C + c) * H + hs + yy) * W + ws + xx];
}

// out[b,c,y,x] = (sum over the tiles covering (y,x), in tile order) / (their count): the reference
// accumulates scoremap * tile with scoremap == 1 and divides by the accumulated score
__global__ __launch_bounds__(256) void merge_tiles_kernel(const float* __restrict__ tiles, float* __restrict__ out,
                                                          TileGrid g, int B, int C, int H, int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)B * C * H * W;
  if (i >= n) return;
  const int xx = (int)(i % W);
  long t = i / W;
  const int yy = (int)(t % H);
  t /= H;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  float sum = 0.f, cnt = 0.f;
  for (int ih = 0; ih < g.nh; ++ih) {
    const int dy = yy - g.hs[ih];
    if (dy < 0 || dy >= g.crop) continue;
    for (int iw = 0; iw < g.nw; ++iw) {
      const int dx = xx - g.ws[iw];
      if (dx < 0 || dx >= g.crop) continue;
      const long tile = (long)ih * g.nw + iw;
      sum += tiles[(((tile * B + b) * C + c) * g.crop + dy) * g.crop + dx];
      cnt += 1.0f;
    }
  }
  out[i] = sum / cnt;
}

// y[b, yy, xx, c] = uint8(rint(clamp(x[b, c, yy, xx], 0, 1) * 255)) for yy < h, xx < w
// (crop to the unpadded size, clamp, NCHW -> NHWC, skimage img_as_ubyte: float32 multiply, round
// half to even)
__global__ __launch_bounds__(256) void crop_clamp_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y,
                                                            int B, int C, int H, int W, int h, int w) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)B * h * w * C;
  if (i >= n) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int xx = (int)(t % w);
  t /= w;
  const int yy = (int)(t % h);
  const int b = (int)(t / h);
  float v = x[(((long)b * C + c) * H + yy) * W + xx];
  v = fminf(fmaxf(v, 0.f), 1.f);
  y[i] = (unsigned char)rintf(v * 255.0f);
}

// same without the byte conversion: y[b,c,yy,xx] = clamp(x[b,c,yy,xx], 0, 1), cropped
__global__ __launch_bounds__(256) void crop_clamp_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         long planes, int H, int W, int h, int w) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = planes * h * w;
  if (i >= n) return;
  const int xx = (int)(i % w);
  const long t = i / w;
  const int yy = (int)(t % h);
  const long pl = t / h;
  y[i] = fminf(fmaxf(x[(pl * H + yy) * W + xx], 0.f), 1.f);
}

static unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

void launch_reflect_pad(const float* x, float* y, int B, int C, int H, int W, int pad_bottom, int pad_right,
                        hipStream_t s) {
  JIRF_REQUIRE(pad_bottom >= 0 && pad_right >= 0 && pad_bottom < H && pad_right < W, JIRF_EINVAL,
               "reflect pad (%d, %d) must be smaller than the image %dx%d", pad_bottom, pad_right, H, W);
  const int Ho = H + pad_bottom, Wo = W + pad_right;
  const long planes = (long)B * C, n = planes * Ho * Wo;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, y, H, W, Ho, Wo, planes);
  JIRF_HIP_CHECK(hipGetLastError());
}

static void check_grid(const TileGrid& g, int H, int W) {
  JIRF_REQUIRE(g.crop > 0 && g.crop <= H && g.crop <= W, JIRF_EINVAL, "tile %d does not fit %dx%d", g.crop, H, W);
  JIRF_REQUIRE(g.nh >= 1 && g.nh <= TILE_GRID_MAX && g.nw >= 1 && g.nw <= TILE_GRID_MAX, JIRF_EINVAL,
               "tile grid %dx%d (at most %d starts per axis)", g.nh, g.nw, TILE_GRID_MAX);
  for (int i = 0; i < g.nh; ++i)
    JIRF_REQUIRE(g.hs[i] >= 0 && g.hs[i] + g.crop <= H, JIRF_EINVAL, "row start %d out of range", g.hs[i]);
  for (int i = 0; i < g.nw; ++i)
    JIRF_REQUIRE(g.ws[i] >= 0 && g.ws[i] + g.crop <= W, JIRF_EINVAL, "column start %d out of range", g.ws[i]);
}

void launch_split_tiles(const float* x, float* tiles, const TileGrid& g, int B, int C, int H, int W, hipStream_t s) {
  check_grid(g, H, W);
  const long n = (long)B * C * g.crop * g.crop * g.nh * g.nw;
  hipLaunchKernelGGL(split_tiles_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, tiles, g, B, C, H, W);
  JIRF_HIP_CHECK(hipGetLastError());
}

void launch_merge_tiles(const float* tiles, float* out, const TileGrid& g, int B, int C, int H, int W, hipStream_t s) {
  check_grid(g, H, W);
  // every pixel must be covered, or the reference's division by the score is 0 / 0
  auto covered = [&](const int* st, int n, int L) {
    std::vector<char> hit(L, 0);
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < g.crop; ++d) hit[st[i] + d] = 1;
    for (char c : hit)
      if (!c) return false;
    return true;
  };
  JIRF_REQUIRE(covered(g.hs, g.nh, H) && covered(g.ws, g.nw, W), JIRF_EINVAL, "the tiles do not cover the image");
  const long n = (long)B * C * H * W;
  hipLaunchKernelGGL(merge_tiles_kernel, dim3(blocks_for(n)), dim3(256), 0, s, tiles, out, g, B, C, H, W);
  JIRF_HIP_CHECK(hipGetLastError());
}

void launch_crop_clamp(const float* x, void* y, int B, int C, int H, int W, int h, int w, bool to_u8, hipStream_t s) {
  JIRF_REQUIRE(h > 0 && w > 0 && h <= H && w <= W, JIRF_EINVAL, "crop %dx%d of %dx%d", h, w, H, W);
  const long n = (long)B * C * h * w;
  if (to_u8)
    hipLaunchKernelGGL(crop_clamp_u8_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, static_cast<unsigned char*>(y),
                       B, C, H, W, h, w);
  else
    hipLaunchKernelGGL(crop_clamp_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, static_cast<float*>(y), (long)B * C,
                       H, W, h, w);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- PSNR statistics (basicsr/metrics/psnr_ssim.py:11-69 of the reference) ------------------------------
// Two planar [C][H][W] images -> sum of squared differences (fp64), maximum of the first image's
// values and the number of values compared, over the window that crop_border leaves.  quantise != 0
// first turns both into the uint8 image tensor2img makes (img_util.py:136-166: clamp to [0, 1], times
// 255 in float32, round half to even); y_channel != 0 then reduces three colour planes to the BT.601
// luma of to_y_channel (metric_util.py:34-47 with matlab_functions.py:268-282): planes scaled to [0, 1]
// in float32, the weighted sum and + 16 in float64, / 255 in float64, stored as float32, times 255 in
// float32.  blue_plane names the plane weighted 24.966 (2 for RGB planes that the reference's
// rgb2bgr swap has turned into BGR, 0 if it was left off and red is taken for blue).
// Deterministic: fixed grid, per-workgroup partials, then one workgroup folds them in index order.
constexpr int PSNR_BLOCKS = 256;

__device__ __forceinline__ float quantise_u8(float v) { return rintf(fminf(fmaxf(v, 0.f), 1.f) * 255.0f); }

__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int C, int H, int W, int border, int quantise,
                                                           int y_channel, int blue_plane, double* __restrict__ work) {
  const int h = H - 2 * border, w = W - 2 * border;
  const long npix = (long)h * w, plane = (long)H * W;
  double sse = 0.0;
  float mx = -INFINITY;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)PSNR_BLOCKS * 256) {
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const long at = (long)(y + border) * W + x + border;
    if (y_channel) {
      double ya = 16.0, yb = 16.0;
      const double wgt[3] = {24.966, 128.553, 65.481};
      for (int k = 0; k < 3; ++k) {                   // k = 0: blue, 1: green, 2: red
        const int pl = k == 1 ? 1 : (k == 0 ? blue_plane : 2 - blue_plane);
        float va = a[pl * plane + at], vb = b[pl * plane + at];
        if (quantise) { va = quantise_u8(va); vb = quantise_u8(vb); }
        ya += (double)(va / 255.0f) * wgt[k];
        yb += (double)(vb / 255.0f) * wgt[k];
      }
      const float fa = (float)(ya / 255.0) * 255.0f, fb = (float)(yb / 255.0) * 255.0f;
      const double d = (double)fa - (double)fb;
      sse += d * d;
      mx = fmaxf(mx, fa);
    } else {
      for (int c = 0; c < C; ++c) {
        float va = a[c * plane + at], vb = b[c * plane + at];
        if (quantise) { va = quantise_u8(va); vb = quantise_u8(vb); }
        const double d = (double)va - (double)vb;
        sse += d * d;
        mx = fmaxf(mx, va);
      }
    }
  }
  __shared__ double s_sse[256];
  __shared__ float s_mx[256];
  s_sse[threadIdx.x] = sse;
  s_mx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_sse[threadIdx.x] += s_sse[threadIdx.x + o];
      s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    work[4 + 2 * blockIdx.x] = s_sse[0];
    work[5 + 2 * blockIdx.x] = (double)s_mx[0];
  }
}

__global__ __launch_bounds__(64) void psnr_final_kernel(double* __restrict__ work, double count) {
  if (threadIdx.x != 0) return;
  double sse = 0.0, mx = -INFINITY;
  for (int i = 0; i < PSNR_BLOCKS; ++i) {
    sse += work[4 + 2 * i];
    mx = fmax(mx, work[5 + 2 * i]);
  }
  work[0] = sse;
  work[1] = mx;
  work[2] = count;
}

size_t psnr_work_doubles() { return 4 + 2 * (size_t)PSNR_BLOCKS; }

void launch_psnr_stats(const float* a, const float* b, int C, int H, int W, int border, int quantise, int y_channel,
                       int blue_plane, double* work, hipStream_t s) {
  JIRF_REQUIRE(border >= 0 && 2 * border < H && 2 * border < W, JIRF_EINVAL, "psnr: crop_border %d leaves nothing of %dx%d",
               border, H, W);
  JIRF_REQUIRE(!y_channel || C == 3, JIRF_EINVAL, "psnr: the Y channel needs 3 colour planes, got %d", C);
  JIRF_REQUIRE(blue_plane == 0 || blue_plane == 2, JIRF_EINVAL, "psnr: blue_plane must be 0 or 2");
  const double count = (double)(H - 2 * border) * (W - 2 * border) * (y_channel ? 1 : C);
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(PSNR_BLOCKS), dim3(256), 0, s, a, b, C, H, W, border, quantise, y_channel,
                     blue_plane, work);
  JIRF_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(psnr_final_kernel, dim3(1), dim3(64), 0, s, work, count);
  JIRF_HIP_CHECK(hipGetLastError());
}

// ---- SSIM statistics (basicsr/metrics/psnr_ssim.py:161-330 of the reference) ----------------------------------
// calculate_ssim has two forms.  Colour images: one 11 x 11 x 11 Gaussian (sigma 1.5, cv2.getGaussianKernel)
// over the H x W x C volume with replicate padding on all three axes (_ssim_3d with _generate_3d_gaussian_kernel:
// the window also runs ACROSS the colour planes), for x, y, x^2, y^2 and x y; test_y_channel: the 2-D window of
// _ssim_cly (cv2.filter2D, BORDER_REPLICATE) on the BT.601 luma.  Both are separable.  The pass over the colour
// axis of a C-plane image with replicate padding is a C x C matrix (host-computed, `mix`), applied while the
// tile is loaded; then 11 taps along x and 11 along y through LDS.  A workgroup owns a 32 x 32 tile of one output
// plane: 42 x 42 inputs of the five quantities.  The colour form computes in fp32 like the reference's convolution
// (_ssim_3d runs on float32 variables); the luma form in double like _ssim_cly (:222-223 cast to float64 before
// cv2.filter2D: on [0, 255] luma the sigma = E[x^2] - mu^2 step cancels about four digits, which fp32 does not
// have to spare against C2 = 58.5), on 16 x 16 tiles so that the double images fit the LDS.  The map is summed in
// double, per workgroup, and the partials are folded in index order (deterministic).
constexpr int SSIM_T = 32, SSIM_TD = 16, SSIM_R = 5;
struct SsimArgs {
  int C, H, W, border, quantise, y_channel, blue_plane, planes_out;
  double c1d, c2d;     // (0.01 max)^2, (0.03 max)^2: rounded to float first for the fp32 form
  double gd[11];       // the window: float values for the fp32 form, doubles for the luma form
  float mix[4][4];
};

template <typename T, int SSIM_T>
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const SsimArgs k, double* __restrict__ part) {
  constexpr int SSIM_IN = SSIM_T + 2 * SSIM_R;
  __shared__ T in[5][SSIM_IN][SSIM_IN + 1];
  __shared__ T tmp[5][SSIM_IN][SSIM_T];
  __shared__ double red[256];
  const int h = k.H - 2 * k.border, w = k.W - 2 * k.border;
  const long plane = (long)k.H * k.W;
  const int x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T, co = blockIdx.z;
  for (int e = threadIdx.x; e < SSIM_IN * SSIM_IN; e += 256) {
    const int iy = e / SSIM_IN, ix = e - iy * SSIM_IN;
    const int yy = min(max(y0 + iy - SSIM_R, 0), h - 1), xx = min(max(x0 + ix - SSIM_R, 0), w - 1);
    const long at = (long)(yy + k.border) * k.W + xx + k.border;
    T q[5] = {0, 0, 0, 0, 0};
    if (k.y_channel) {
      double ya = 16.0, yb = 16.0;
      const double wgt[3] = {24.966, 128.553, 65.481};
      for (int c = 0; c < 3; ++c) {                   // c = 0: blue, 1: green, 2: red
        const int pl = c == 1 ? 1 : (c == 0 ? k.blue_plane : 2 - k.blue_plane);
        float va = a[pl * plane + at], vb = b[pl * plane + at];
        if (k.quantise) { va = quantise_u8(va); vb = quantise_u8(vb); }
        ya += (double)(va / 255.0f) * wgt[c];
        yb += (double)(vb / 255.0f) * wgt[c];
      }
      // to_y_channel returns float32 luma; _ssim_cly widens it to float64 before anything is multiplied
      const T fa = (T)((float)(ya / 255.0) * 255.0f), fb = (T)((float)(yb / 255.0) * 255.0f);
      q[0] = fa; q[1] = fb; q[2] = fa * fa; q[3] = fb * fb; q[4] = fa * fb;
    } else {
      for (int c = 0; c < k.C; ++c) {
        const float m = k.mix[co][c];
        if (m == 0.f) continue;
        float va = a[c * plane + at], vb = b[c * plane + at];
        if (k.quantise) { va = quantise_u8(va); vb = quantise_u8(vb); }
        q[0] += (T)(m * va); q[1] += (T)(m * vb);
        q[2] += (T)(m * (va * va)); q[3] += (T)(m * (vb * vb)); q[4] += (T)(m * (va * vb));
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) in[j][iy][ix] = q[j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < SSIM_IN * SSIM_T; e += 256) {
    const int iy = e / SSIM_T, ox = e - iy * SSIM_T;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      T t = 0;
#pragma unroll
      for (int u = 0; u < 11; ++u) t += (T)k.gd[u] * in[j][iy][ox + u];
      tmp[j][iy][ox] = t;
    }
  }
  __syncthreads();
  double acc = 0.0;
  for (int e = threadIdx.x; e < SSIM_T * SSIM_T; e += 256) {
    const int oy = e / SSIM_T, ox = e - oy * SSIM_T;
    if (y0 + oy >= h || x0 + ox >= w) continue;
    T f[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      T t = 0;
#pragma unroll
      for (int u = 0; u < 11; ++u) t += (T)k.gd[u] * tmp[j][oy + u][ox];
      f[j] = t;
    }
    const T c1 = (T)k.c1d, c2 = (T)k.c2d;
    const T mu1 = f[0], mu2 = f[1];
    const T mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const T s1 = f[2] - mu1_sq, s2 = f[3] - mu2_sq, s12 = f[4] - mu12;
    acc += (double)((((T)2 * mu12 + c1) * ((T)2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2)));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[4 + ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void ssim_final_kernel(double* __restrict__ work, long n_part, double count) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (long i = 0; i < n_part; ++i) t += work[4 + i];
  work[0] = t;
  work[1] = count;
}

size_t ssim_work_doubles(int C, int H, int W) {
  // enough for either tiling (the luma form has one output plane of 16 x 16 tiles)
  return 4 + std::max((size_t)std::max(C, 1) * ((H + SSIM_T - 1) / SSIM_T) * ((W + SSIM_T - 1) / SSIM_T),
                      (size_t)((H + SSIM_TD - 1) / SSIM_TD) * ((W + SSIM_TD - 1) / SSIM_TD));
}

void launch_ssim_stats(const float* a, const float* b, int C, int H, int W, int border, int quantise, int y_channel,
                       int blue_plane, float max_value, double* work, hipStream_t s) {
  JIRF_REQUIRE(border >= 0 && 2 * border < H && 2 * border < W, JIRF_EINVAL, "ssim: crop_border %d leaves nothing of %dx%d",
               border, H, W);
  JIRF_REQUIRE(C >= 1 && C <= 4, JIRF_EINVAL, "ssim: %d planes (1 to 4 are built)", C);
  JIRF_REQUIRE(!y_channel || C == 3, JIRF_EINVAL, "ssim: the Y channel needs 3 colour planes, got %d", C);
  JIRF_REQUIRE(blue_plane == 0 || blue_plane == 2, JIRF_EINVAL, "ssim: blue_plane must be 0 or 2");
  SsimArgs k{};
  k.C = C; k.H = H; k.W = W; k.border = border; k.quantise = quantise; k.y_channel = y_channel; k.blue_plane = blue_plane;
  k.planes_out = y_channel ? 1 : C;
  if (y_channel) {
    k.c1d = (0.01 * (double)max_value) * (0.01 * (double)max_value);
    k.c2d = (0.03 * (double)max_value) * (0.03 * (double)max_value);
  } else {
    k.c1d = (double)((0.01f * max_value) * (0.01f * max_value));
    k.c2d = (double)((0.03f * max_value) * (0.03f * max_value));
  }
  // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1
  double g[11], gs = 0.0;
  for (int i = 0; i < 11; ++i) { g[i] = std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); gs += g[i]; }
  for (int i = 0; i < 11; ++i) k.gd[i] = y_channel ? g[i] / gs : (double)(float)(g[i] / gs);
  for (int co = 0; co < C; ++co)
    for (int t = 0; t < 11; ++t) k.mix[co][std::min(std::max(co + t - 5, 0), C - 1)] += (float)(g[t] / gs);
  const int h = H - 2 * border, w = W - 2 * border;
  const int tile = y_channel ? SSIM_TD : SSIM_T;
  const dim3 grid((w + tile - 1) / tile, (h + tile - 1) / tile, k.planes_out);
  if (y_channel)
    hipLaunchKernelGGL((ssim_partial_kernel<double, SSIM_TD>), grid, dim3(256), 0, s, a, b, k, work);
  else
    hipLaunchKernelGGL((ssim_partial_kernel<float, SSIM_T>), grid, dim3(256), 0, s, a, b, k, work);
  JIRF_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(64), 0, s, work, (long)grid.x * grid.y * grid.z,
                     (double)h * w * k.planes_out);
  JIRF_HIP_CHECK(hipGetLastError());
}

}  // namespace jirf
