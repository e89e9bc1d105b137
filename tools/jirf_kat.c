/* Plain-C client of libjirf_hip.so: runs one known-answer case through the C ABI alone (no Python, no
 * HIP calls of its own: inputs and outputs are host buffers) and reports the largest deviation from
 * the expected output stored in the case file.  Shows the call sequence a non-Python caller uses:
 *   jirf_net_create -> jirf_net_set_param (per state-dict entry) -> jirf_net_finalize -> jirf_net_plan
 *   -> [jirf_net_set_forced_perms] -> jirf_net_forward.
 *
 * Case file (little endian), written by tests/test_c_client_gpu.py:
 *   "JKAT1\0\0\0"                                   8 bytes
 *   u32 len + bytes: arch name;  u32 len + bytes: network options as JSON
 *   u32 batch_mode, u32 B, u32 H, u32 W, u32 C_in, u32 C_out
 *   f32 tolerance
 *   u32 n_params, then per parameter: u32 len + key, u32 ndim, i64 shape[ndim], f32 data[prod(shape)]
 *   u64 n_perm_ints + i32 perms[n]                  (0: channel orders are not pinned)
 *   f32 x[B*C_in*H*W], f32 expected[B*C_out*H*W]
 * Exit status 0: max |y - expected| <= tolerance.
 *
 * Build: gcc -O2 -std=c99 -Iinclude tools/jirf_kat.c -Limage-restoration_amd -ljirf_hip \
 *            -Wl,-rpath,'$ORIGIN/../image-restoration_amd' -lm -o tools/jirf_kat */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jirf_hip.h"

static void die(const char* what) {
  fprintf(stderr, "jirf_kat: %s: %s\n", what, jirf_last_error());
  exit(2);
}

static void rd(FILE* f, void* dst, size_t n) {
  if (fread(dst, 1, n, f) != n) {
    fprintf(stderr, "jirf_kat: case file is truncated\n");
    exit(2);
  }
}

static uint32_t rd_u32(FILE* f) {
  uint32_t v;
  rd(f, &v, 4);
  return v;
}

static char* rd_str(FILE* f) {
  const uint32_t n = rd_u32(f);
  char* s = (char*)malloc((size_t)n + 1);
  if (!s) exit(2);
  rd(f, s, n);
  s[n] = 0;
  return s;
}

int main(int argc, char** argv) {
  if (argc != 2) {
    fprintf(stderr, "usage: jirf_kat <case file>\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  char magic[8];
  rd(f, magic, 8);
  if (memcmp(magic, "JKAT1\0\0\0", 8) != 0) {
    fprintf(stderr, "jirf_kat: not a case file\n");
    return 2;
  }
  if (jirf_abi_version() != JIRF_ABI_VERSION) {
    fprintf(stderr, "jirf_kat: header is ABI %d, library is ABI %d\n", JIRF_ABI_VERSION, jirf_abi_version());
    return 2;
  }
  char* arch = rd_str(f);
  char* cfg = rd_str(f);
  const int mode = (int)rd_u32(f), B = (int)rd_u32(f), H = (int)rd_u32(f), W = (int)rd_u32(f);
  const int Cin = (int)rd_u32(f), Cout = (int)rd_u32(f);
  float tol;
  rd(f, &tol, 4);

  jirf_net* net = jirf_net_create(arch, cfg, 0);
  if (!net) die("jirf_net_create");
  const uint32_t n_params = rd_u32(f);
  for (uint32_t i = 0; i < n_params; ++i) {
    char* key = rd_str(f);
    const uint32_t ndim = rd_u32(f);
    int64_t shape[8];
    size_t n = 1;
    if (ndim > 8) {
      fprintf(stderr, "jirf_kat: %s has %u dims\n", key, ndim);
      return 2;
    }
    for (uint32_t d = 0; d < ndim; ++d) {
      rd(f, &shape[d], 8);
      n *= (size_t)shape[d];
    }
    float* data = (float*)malloc(n * sizeof(float) + 4);
    if (!data) return 2;
    rd(f, data, n * sizeof(float));
    if (jirf_net_set_param(net, key, data, JIRF_F32, shape, (int)ndim) != JIRF_OK) die(key);
    free(data);
    free(key);
  }
  if (jirf_net_finalize(net) != JIRF_OK) die("jirf_net_finalize");
  size_t arena = 0;
  if (jirf_net_plan(net, B, H, W, mode, &arena) != JIRF_OK) die("jirf_net_plan");

  uint64_t n_perm;
  rd(f, &n_perm, 8);
  if (n_perm) {
    int32_t* perms = (int32_t*)malloc((size_t)n_perm * 4);
    if (!perms) return 2;
    rd(f, perms, (size_t)n_perm * 4);
    if (jirf_net_set_forced_perms(net, perms, (size_t)n_perm) != JIRF_OK) die("jirf_net_set_forced_perms");
    free(perms);
  }
  const size_t nx = (size_t)B * Cin * H * W, ny = (size_t)B * Cout * H * W;
  float* x = (float*)malloc(nx * sizeof(float));
  float* want = (float*)malloc(ny * sizeof(float));
  float* y = (float*)malloc(ny * sizeof(float));
  if (!x || !want || !y) return 2;
  rd(f, x, nx * sizeof(float));
  rd(f, want, ny * sizeof(float));
  fclose(f);

  if (jirf_net_forward(net, x, y, B, H, W, mode, /*on_device=*/0, NULL) != JIRF_OK) die("jirf_net_forward");
  double worst = 0.0;
  for (size_t i = 0; i < ny; ++i) {
    const double d = fabs((double)y[i] - (double)want[i]);
    if (!(d <= worst)) worst = d;   /* also catches NaN */
  }
  printf("%s %dx%dx%dx%d arena %zu bytes  max_abs_err %.3e  tolerance %.1e\n", arch, B, Cin, H, W, arena, worst, (double)tol);
  jirf_net_destroy(net);
  free(x); free(want); free(y); free(arch); free(cfg);
  return worst <= (double)tol ? 0 : 1;
}
