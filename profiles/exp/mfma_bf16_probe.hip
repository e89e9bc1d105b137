// What the bf16 matrix instruction of the split GEMM (v_mfma_f32_32x32x16_bf16) sustains on one MI355X under load:
// bare register-operand loops, four accumulators per wave, 1 to 4 waves per SIMD, with random (non-zero) operands.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_probe mfma_bf16_probe.hip && ./mfma_bf16_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(e)                                                       \
  do {                                                              \
    hipError_t _e = (e);                                            \
    if (_e != hipSuccess) {                                         \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));       \
      exit(1);                                                      \
    }                                                               \
  } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)in[(tid * 32 + i * 8 + j) & 1023];
      b[i][j] = (__bf16)in[(tid * 32 + i * 8 + j + 517) & 1023];
    }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) & 3], acc[j % NACC], 0, 0, 0);
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][15];
  out[blockIdx.x * 256 + tid] = s;
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
  const int iters = 8000;
  for (int nacc = 1; nacc <= 4; nacc += 3)
    for (int wps = 1; wps <= 4; ++wps) {
      const int blocks = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (nacc == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
        if (rep)
          printf("v_mfma_f32_32x32x16_bf16  %d accumulator(s), %d wave(s) per SIMD: %.2f ms  %.0f TFLOP/s  (%.2f GHz-equivalent of 1024 flop/cycle/SIMD)\n",
                 nacc, wps, ms, flop / ms / 1e9, flop / ms / 1e9 / (1024.0 * 1024 / 1e3));
      }
    }
  return 0;
}
