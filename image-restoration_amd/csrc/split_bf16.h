// Device-side pieces shared by the split-bf16 GEMM kernels: an fp32 value as three bf16 terms.
#pragma once
#include <hip/hip_runtime.h>

namespace jirf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// eight fp32 values -> hi, mid, lo with hi + mid + lo == v exactly (8 + 8 + 8 significand bits, round to nearest
// at each step, the remainders are exact), each packed as the eight consecutive k of one MFMA operand
__device__ __forceinline__ void split8(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hi = (__bf16)v[i];
    const float r1 = v[i] - (float)hi;
    const __bf16 mi = (__bf16)r1;
    const float r2 = r1 - (float)mi;
    h[i] = hi;
    m[i] = mi;
    l[i] = (__bf16)r2;
  }
}

}  // namespace jirf
