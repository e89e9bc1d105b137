// Host runtime of libjirf_hip: parameter registry, config parsing, device buffers, kernel clock.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace jirf {

// ---- tiny JSON reader (objects of scalars / flat arrays — what a network_g YAML block holds) ----
struct JsonValue {
  enum Kind { NUL, BOOL, NUM, STR, ARR } kind = NUL;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<double> arr;
};
using JsonObject = std::map<std::string, JsonValue>;
JsonObject parse_json_object(const char* text);

struct Config {
  JsonObject obj;
  bool has(const std::string& k) const { return obj.count(k) != 0; }
  int get_int(const std::string& k, int dflt) const;
  double get_num(const std::string& k, double dflt) const;
  bool get_bool(const std::string& k, bool dflt) const;
  std::string get_str(const std::string& k, const std::string& dflt) const;
  std::vector<int> get_ints(const std::string& k, const std::vector<int>& dflt) const;
};

// ---- device memory ---------------------------------------------------------------------------
struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  void ensure(size_t n);  // grows (never shrinks); contents are lost on growth
  void release();
};

// Bump allocator over one DeviceBuffer; `measure` mode only counts.
struct Arena {
  unsigned char* base = nullptr;
  size_t off = 0;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  }
  float* floats(size_t n) { return static_cast<float*>(take(n * sizeof(float))); }
};

// ---- kernel clock -----------------------------------------------------------------------------
struct KernelClock {
  struct Rec {
    const char* name;
    double flops, bytes;
    hipEvent_t e0, e1;
  };
  bool enabled = false;
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  // aggregated result of the last collect()
  struct Agg {
    const char* name;
    float ms;
    double flops, bytes;
    int launches;
  };
  std::vector<Agg> agg;
  hipEvent_t get_event();
  void begin_forward();
  void collect();  // synchronises the recorded events and aggregates by name
  ~KernelClock();
};

// ---- parameters --------------------------------------------------------------------------------
struct ParamSpec {
  std::string key;
  std::vector<int64_t> shape;
  std::vector<float> host;  // reference layout, fp32
  bool is_set = false;
  bool optional = false;    // buffers the reference recomputes (e.g. index tables)
  size_t numel() const {
    size_t n = 1;
    for (auto d : shape) n *= (size_t)d;
    return n;
  }
};

struct Net {
  std::string arch;
  int device = 0;
  Config cfg;
  std::vector<ParamSpec> params;
  std::unordered_map<std::string, int> index;
  bool finalized = false;
  KernelClock clock;
  // set by plan(): device staging for host-pointer forwards and the I/O channel counts
  float* stage_in = nullptr;
  float* stage_out = nullptr;
  int io_in_ch = 0, io_out_ch = 0;

  virtual ~Net() = default;
  void add_param(const std::string& key, std::vector<int64_t> shape, bool optional = false);
  const std::vector<float>& host(const std::string& key) const;
  bool has_param(const std::string& key) const { return index.count(key) != 0; }
  // maps aliases / prefixes of the reference's checkpoints onto registered keys; returns ""
  // for keys that are accepted and ignored
  virtual std::string canonical_key(const std::string& key) const;
  int set_param(const char* key, const void* data, int dtype, const int64_t* shape, int ndim);

  virtual void finalize() = 0;
  virtual size_t plan(int B, int H, int W, int mode) = 0;
  virtual void forward(const float* x, float* y, int B, int H, int W, int mode, hipStream_t s) = 0;
  // HINT only; defaults report "not supported"
  virtual void reset_cache() {}
  virtual int get_cache(float*, int*, int*, size_t) { return JIRF_ESTATE; }
  virtual int set_cache(const float*, int, int) { return JIRF_ESTATE; }
  virtual int attention_blocks(int*, int*, int) { return JIRF_ESTATE; }
  virtual int get_perms(int32_t*, size_t, size_t*) { return JIRF_ESTATE; }
  virtual int set_forced_perms(const int32_t*, size_t) { return JIRF_ESTATE; }
};

std::unique_ptr<Net> make_hint_net(const Config& cfg);
std::unique_ptr<Net> make_astv2_net(const Config& cfg);
std::unique_ptr<Net> make_ast_net(const Config& cfg);
std::unique_ptr<Net> make_fpro_net(const Config& cfg);

// Uploads a host vector into a weight pool that lives as long as the net.
struct WeightPool {
  std::vector<std::unique_ptr<DeviceBuffer>> bufs;
  const float* upload(const std::vector<float>& v);
  const float* upload(const float* v, size_t n);
};

struct PackedW {
  const float* wt = nullptr;
  const uint16_t* wt3 = nullptr;   // split-bf16 planes (only when that GEMM variant is enabled)
  long wt3_plane = 0;
  const float* bias = nullptr;
  int M = 0, K = 0, Mpad = 0, Cin = 0;
};
PackedW pack_upload_1x1(WeightPool& pool, const float* w, const float* bias, int M, int K);
// A 1x1 convolution that follows a LayerNorm with weight lnw [K] and bias lnb [K] (or null):
//   W (n * lnw + lnb) + bias  =  (W diag(lnw)) n + (W lnb + bias)        (n = the normalised input)
// so the packed weights carry lnw and the packed bias W lnb (summed in double); the GEMM's LayerNorm
// loader is then called with lnw = lnb = null and only normalises.
PackedW pack_upload_ln_1x1(WeightPool& pool, const float* w, const float* bias, const float* lnw, const float* lnb,
                           int M, int K);
PackedW pack_upload_3x3(WeightPool& pool, const float* w, const float* bias, int M, int Cin);
PackedW pack_upload_conv(WeightPool& pool, const float* w, const float* bias, int M, int Cin, int ks);


// ---- the gated feed-forward block as one kernel (gdfn_fused.hip) -------------------------------------------------
struct GdfnFusedW {          // device pointers into a WeightPool, in the kernel's lane-major packings
  const float *win = nullptr, *bin = nullptr, *dwp = nullptr, *dwb = nullptr, *wout = nullptr, *bout = nullptr;
  int C = 0, hidden = 0;
};
// true when launch_gdfn_fused takes the shape (channels, hidden width, plane size)
bool gdfn_fused_supported(int C, int hidden, int H, int W);
// Host arrays in the reference's layouts: LayerNorm weight / bias [C] (bias null: BiasFree), project_in [2 hidden][C],
// dwconv [2 hidden][9], project_out [C][hidden], their biases or null.
GdfnFusedW pack_upload_gdfn_fused(WeightPool& pool, int C, int hidden, const float* ln_w, const float* ln_b,
                                  const float* w_in, const float* b_in, const float* w_dw, const float* b_dw,
                                  const float* w_out, const float* b_out);
// y = x + project_out(gelu(dw(T)[:h]) * dw(T)[h:]), T = project_in(LayerNorm(x)); x and y must not alias.
void launch_gdfn_fused(const GdfnFusedW& w, TV x, TV y, int B, int H, int W, bool ln_center, hipStream_t s);

}  // namespace jirf
