// What the fp32 matrix pipe of one MI355X sustains, and at which clock: bare v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 loops on random operands (registers only), and the same loop with its B
// operand re-read from LDS one step or several steps ahead.  Prints TFLOP/s and the in-kernel shader
// clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_probe mfma_f32_probe.hip && ./mfma_f32_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Stamp {
  unsigned long long cyc, real;
};

// MODE 0: 32x32x2, operands in registers, NACC accumulators
// MODE 1: 16x16x4, operands in registers, NACC accumulators
// MODE 2: 32x32x2, B from LDS, read right before use (depth 0)
// MODE 3: 32x32x2, B from LDS, read one group of NACC ahead
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters,
                                              Stamp* stamps) {
  __shared__ float lds[64 * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 128; i += 256) lds[i] = in[(blockIdx.x * 97 + i) & 0xffff];
  __syncthreads();
  float a[8], bq[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    bq[i] = in[(tid * 8 + i + 4099) & 0xffff];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE == 1) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[(i + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* bp = lds + (lane & 31) + (lane >> 5) * 128;
    if constexpr (MODE == 0) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NACC; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[(i + j) & 7], acc[j], 0, 0, 0);
      }
    } else if constexpr (MODE == 2) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NACC; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bp[(2 * i) * 128 + j * 32 + (it & 3) * 2048], acc[j], 0, 0, 0);
      }
    } else {
      float bn[NACC], bc[NACC];
#pragma unroll
      for (int j = 0; j < NACC; ++j) bc[j] = bp[j * 32];
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < NACC; ++j) bn[j] = bp[(2 * ((i + 1) & 7)) * 128 + j * 32 + (it & 3) * 2048];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bc[j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NACC; ++j) bc[j] = bn[j];
        }
      }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][7];
    out[blockIdx.x * 256 + tid] = s;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

// The first n_mfma_wg x 256 workgroups (n_mfma_wg per CU) run an MFMA loop (operands in registers, or B
// re-read from LDS), the last 256 a loop of INDEPENDENT plain vector instructions (8 chains of
// v_fma_f32), and each kind reports its own cycles: what a vector instruction costs a wave whose SIMD
// is shared with waves that always have an MFMA queued.
__global__ __launch_bounds__(256) void mixed(const float* __restrict__ in, float* __restrict__ out, int iters,
                                             Stamp* stamps, int n_mfma_wg, int b_from_lds) {
  __shared__ float lds[64 * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const bool mfma_wg = (int)blockIdx.x < 256 * n_mfma_wg;
  for (int i = tid; i < 64 * 128; i += 256) lds[i] = in[(blockIdx.x * 97 + i) & 0xffff];
  __syncthreads();
  float a[8], bq[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    bq[i] = in[(tid * 8 + i + 4099) & 0xffff];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (mfma_wg) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (b_from_lds) {
      const float* bp = lds + 4 * (lane & 31) + (lane >> 5) * 128;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bp[(2 * i) * 128 + (it & 3) * 2048]);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b4[j], acc[j], 0, 0, 0);
        }
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[(i + j) & 7], acc[j], 0, 0, 0);
      }
    }
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  } else {
    // iters / 4 trips x 64 v_fma: a quarter of an MFMA workgroup's instruction count
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], bq[i], bq[(i + u) & 7]);
    }
    for (int i = 0; i < 8; ++i) s += a[i];
  }
  out[blockIdx.x * 256 + tid] = s;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

// Every wave runs the MFMA loop AND, between its own MFMAs, NV independent plain vector instructions per k-step
// (4 MFMAs): what vector work costs when it sits in the issuing wave's own MFMA shadows, with one or two such
// waves per SIMD.
template <int NV>
__global__ __launch_bounds__(256) void interleaved(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                   Stamp* stamps) {
  const int tid = threadIdx.x;
  float a[8], bq[8], v[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    bq[i] = in[(tid * 8 + i + 4099) & 0xffff];
    v[i] = in[(tid * 8 + i + 977) & 0xffff];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[(i + j) & 7], acc[j], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NV; ++u) v[u & 7] = fmaf(v[u & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + tid] = s;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

template <int NV>
void run_interleaved(const float* in, float* out, Stamp* st, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(interleaved<NV>, dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
  CK(hipDeviceSynchronize());
  std::vector<Stamp> h(blocks);
  CK(hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
  std::vector<double> m;
  for (int i = 0; i < blocks; ++i) m.push_back((double)h[i].cyc);
  std::sort(m.begin(), m.end());
  printf("%d wave(s) per SIMD, %d vector instructions between every 4 MFMAs of the same wave: %.1f cycles per MFMA per wave "
         "(pipe-bound: %d)\n", wg_per_cu, NV, m[m.size() / 2] / (iters * 32.0), 64 * wg_per_cu);
}

void run_mixed(const float* in, float* out, Stamp* st, int n_mfma_wg, int b_from_lds) {
  const int iters = 4000, blocks = 256 * (n_mfma_wg + 1);
  for (int rep = 0; rep < 10; ++rep)
    hipLaunchKernelGGL(mixed, dim3(blocks), dim3(256), 0, 0, in, out, iters, st, n_mfma_wg, b_from_lds);
  CK(hipDeviceSynchronize());
  std::vector<Stamp> h(blocks);
  CK(hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
  std::vector<double> m, v;
  for (int i = 0; i < 256 * n_mfma_wg; ++i) m.push_back((double)h[i].cyc);
  for (int i = 256 * n_mfma_wg; i < blocks; ++i) v.push_back((double)h[i].cyc);
  std::sort(m.begin(), m.end());
  std::sort(v.begin(), v.end());
  const double n_valu = iters / 4 * 64.0, n_mfma = iters * 32.0;
  if (n_mfma_wg == 0)
    printf("vector-only workgroup (1 per CU): %.1f cycles per v_fma_f32 per wave\n", v[128] / n_valu);
  else
    printf("%d MFMA workgroup(s) (B from %s) + 1 vector workgroup per CU: %.1f cycles per MFMA per wave, %.1f cycles per v_fma_f32 beside them\n",
           n_mfma_wg, b_from_lds ? "LDS" : "registers", m[m.size() / 2] / n_mfma, v[128] / n_valu);
}

template <int MODE, int NACC>
void run(const char* what, int wg_per_cu, const float* in, float* out, Stamp* st) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0.f;
  // >= 2 s of back-to-back launches so the clock settles, the last one is timed
  for (int rep = 0; rep < 3; ++rep) {
    const int n = rep < 2 ? 20 : 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((probe<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, in, out, iters, st);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
  }
  std::vector<Stamp> h(blocks);
  CK(hipMemcpy(h.data(), st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  for (auto& s : h) mhz.push_back((double)s.cyc / (double)s.real * 100.0);
  std::sort(mhz.begin(), mhz.end());
  const double per = MODE == 1 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
  const double flops = per * 8.0 * NACC * iters * 4.0 * blocks;
  printf("%-58s %d WG/CU  %8.3f ms  %7.1f TFLOP/s  clock %6.0f MHz (median)\n", what, wg_per_cu, ms, flops / ms / 1e9,
         mhz[mhz.size() / 2]);
}

int main() {
  float *in, *out;
  Stamp* st;
  CK(hipMalloc(&in, 65536 * 4));
  CK(hipMalloc(&out, 1024 * 256 * 4));
  CK(hipMalloc(&st, 1024 * sizeof(Stamp)));
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  run<0, 4>("32x32x2 registers, 4 accumulators", 1, in, out, st);
  run<0, 4>("32x32x2 registers, 4 accumulators", 2, in, out, st);
  run<0, 1>("32x32x2 registers, 1 accumulator", 1, in, out, st);
  run<1, 4>("16x16x4 registers, 4 accumulators", 1, in, out, st);
  run<1, 4>("16x16x4 registers, 4 accumulators", 2, in, out, st);
  run<2, 4>("32x32x2, B from LDS as the compiler schedules it", 1, in, out, st);
  run<2, 4>("32x32x2, B from LDS as the compiler schedules it", 2, in, out, st);
  run<3, 4>("32x32x2, B from LDS one k-step ahead", 1, in, out, st);
  run<3, 4>("32x32x2, B from LDS one k-step ahead", 2, in, out, st);
  run_interleaved<0>(in, out, st, 1);
  run_interleaved<2>(in, out, st, 1);
  run_interleaved<4>(in, out, st, 1);
  run_interleaved<8>(in, out, st, 1);
  run_interleaved<16>(in, out, st, 1);
  run_interleaved<0>(in, out, st, 2);
  run_interleaved<2>(in, out, st, 2);
  run_interleaved<4>(in, out, st, 2);
  run_interleaved<8>(in, out, st, 2);
  run_interleaved<16>(in, out, st, 2);
  run_mixed(in, out, st, 0, 0);
  run_mixed(in, out, st, 1, 0);
  run_mixed(in, out, st, 2, 0);
  run_mixed(in, out, st, 1, 1);
  run_mixed(in, out, st, 2, 1);
  return 0;
}
