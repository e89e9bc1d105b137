// What the fp64 matrix instruction hint_chan_stats uses (v_mfma_f64_16x16x4_f64) sustains on one MI355X next to the
// fp32 one (v_mfma_f32_32x32x2_f32): bare register-operand loops, four accumulators per wave, four waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define CK(e)                                                       \
  do {                                                              \
    hipError_t _e = (e);                                            \
    if (_e != hipSuccess) {                                         \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));       \
      exit(1);                                                      \
    }                                                               \
  } while (0)

template <bool F64>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 4 + i) & 1023];
    b[i] = in[(tid * 4 + i + 517) & 1023];
  }
  if constexpr (F64) {
    f64x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[i], (double)b[(i + j) & 3], acc[j], 0, 0, 0);
    double s = 0.0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + tid] = (float)s;
  } else {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + j) & 3], acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
    out[blockIdx.x * 256 + tid] = s;
  }
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 1024 * 4));
  CK(hipMalloc(&out, 4096 * 256 * 4));
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = 256 * 4, iters = 4000;
  for (int f64 = 0; f64 < 2; ++f64) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (f64) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double mfma = (double)blocks * 4 * iters * 16;            // per wave: iters x 16 instructions
      const double flop = mfma * (f64 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2);
      if (rep) printf("%s  %.2f ms  %.1f TFLOP/s\n", f64 ? "v_mfma_f64_16x16x4_f64 " : "v_mfma_f32_32x32x2_f32", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
