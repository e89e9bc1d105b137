// Host runtime of libjirf_hip: errors, JSON config, device buffers, kernel clock, parameter registry.
#include "runtime.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace jirf {

// ---- errors ------------------------------------------------------------------------------------
static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}
const char* last_error() { return g_error.c_str(); }

// ---- JSON ----------------------------------------------------------------------------------------
namespace {
struct JsonReader {
  const char* s;
  void ws() {
    while (*s && std::isspace((unsigned char)*s)) ++s;
  }
  [[noreturn]] void fail(const char* what) {
    set_error("config JSON: %s near '%.20s'", what, s);
    throw Failure{JIRF_EINVAL};
  }
  std::string string() {
    if (*s != '"') fail("expected string");
    ++s;
    std::string out;
    while (*s && *s != '"') {
      if (*s == '\\' && s[1]) ++s;
      out.push_back(*s++);
    }
    if (*s != '"') fail("unterminated string");
    ++s;
    return out;
  }
  double number() {
    char* end = nullptr;
    const double v = std::strtod(s, &end);
    if (end == s) fail("expected number");
    s = end;
    return v;
  }
  JsonValue value() {
    ws();
    JsonValue v;
    if (*s == '"') {
      v.kind = JsonValue::STR;
      v.str = string();
    } else if (*s == '[') {
      v.kind = JsonValue::ARR;
      ++s;
      ws();
      while (*s && *s != ']') {
        ws();
        if (!std::strncmp(s, "true", 4)) { v.arr.push_back(1); s += 4; }
        else if (!std::strncmp(s, "false", 5)) { v.arr.push_back(0); s += 5; }
        else v.arr.push_back(number());
        ws();
        if (*s == ',') ++s;
        ws();
      }
      if (*s != ']') fail("unterminated array");
      ++s;
    } else if (!std::strncmp(s, "true", 4)) {
      v.kind = JsonValue::BOOL; v.b = true; s += 4;
    } else if (!std::strncmp(s, "false", 5)) {
      v.kind = JsonValue::BOOL; v.b = false; s += 5;
    } else if (!std::strncmp(s, "null", 4)) {
      v.kind = JsonValue::NUL; s += 4;
    } else {
      v.kind = JsonValue::NUM;
      v.num = number();
    }
    return v;
  }
};
}  // namespace

JsonObject parse_json_object(const char* text) {
  JsonObject out;
  if (!text) return out;
  JsonReader r{text};
  r.ws();
  if (!*r.s) return out;
  if (*r.s != '{') r.fail("expected object");
  ++r.s;
  r.ws();
  while (*r.s && *r.s != '}') {
    r.ws();
    std::string key = r.string();
    r.ws();
    if (*r.s != ':') r.fail("expected ':'");
    ++r.s;
    out[key] = r.value();
    r.ws();
    if (*r.s == ',') ++r.s;
    r.ws();
  }
  if (*r.s != '}') r.fail("unterminated object");
  return out;
}

int Config::get_int(const std::string& k, int dflt) const {
  auto it = obj.find(k);
  if (it == obj.end() || it->second.kind == JsonValue::NUL) return dflt;
  if (it->second.kind == JsonValue::BOOL) return it->second.b;
  JIRF_REQUIRE(it->second.kind == JsonValue::NUM, JIRF_EINVAL, "config key '%s' must be a number", k.c_str());
  return (int)it->second.num;
}
double Config::get_num(const std::string& k, double dflt) const {
  auto it = obj.find(k);
  if (it == obj.end() || it->second.kind == JsonValue::NUL) return dflt;
  JIRF_REQUIRE(it->second.kind == JsonValue::NUM, JIRF_EINVAL, "config key '%s' must be a number", k.c_str());
  return it->second.num;
}
bool Config::get_bool(const std::string& k, bool dflt) const {
  auto it = obj.find(k);
  if (it == obj.end() || it->second.kind == JsonValue::NUL) return dflt;
  if (it->second.kind == JsonValue::NUM) return it->second.num != 0.0;
  JIRF_REQUIRE(it->second.kind == JsonValue::BOOL, JIRF_EINVAL, "config key '%s' must be a bool", k.c_str());
  return it->second.b;
}
std::string Config::get_str(const std::string& k, const std::string& dflt) const {
  auto it = obj.find(k);
  if (it == obj.end() || it->second.kind == JsonValue::NUL) return dflt;
  JIRF_REQUIRE(it->second.kind == JsonValue::STR, JIRF_EINVAL, "config key '%s' must be a string", k.c_str());
  return it->second.str;
}
std::vector<int> Config::get_ints(const std::string& k, const std::vector<int>& dflt) const {
  auto it = obj.find(k);
  if (it == obj.end() || it->second.kind == JsonValue::NUL) return dflt;
  JIRF_REQUIRE(it->second.kind == JsonValue::ARR, JIRF_EINVAL, "config key '%s' must be an array", k.c_str());
  std::vector<int> out;
  for (double d : it->second.arr) out.push_back((int)d);
  return out;
}

// ---- device memory ---------------------------------------------------------------------------
void DeviceBuffer::ensure(size_t n) {
  if (n <= bytes) return;
  release();
  hipError_t e = hipMalloc(&p, n);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc of %zu bytes failed: %s", n, hipGetErrorString(e));
    throw Failure{JIRF_ENOMEM};
  }
  bytes = n;
}
void DeviceBuffer::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

const float* WeightPool::upload(const float* v, size_t n) {
  auto buf = std::make_unique<DeviceBuffer>();
  buf->ensure(std::max<size_t>(n, 1) * sizeof(float));
  if (n) JIRF_HIP_CHECK(hipMemcpy(buf->p, v, n * sizeof(float), hipMemcpyHostToDevice));
  const float* p = static_cast<const float*>(buf->p);
  bufs.push_back(std::move(buf));
  return p;
}
const float* WeightPool::upload(const std::vector<float>& v) { return upload(v.data(), v.size()); }

PackedW pack_upload_1x1(WeightPool& pool, const float* w, const float* bias, int M, int K) {
  std::vector<float> packed;
  PackedW out;
  pack_wt_1x1(w, M, K, packed, &out.Mpad);
  out.wt = pool.upload(packed);
  out.bias = bias ? pool.upload(bias, M) : nullptr;
  out.M = M; out.K = K; out.Cin = K;
  if (gemm_split_bf16_enabled()) {
    std::vector<uint16_t> planes;
    out.wt3_plane = (long)pack_wt_split3(w, M, K, out.Mpad, planes);
    // uploaded through the float pool: two bf16 per float slot
    planes.resize((planes.size() + 1) / 2 * 2, 0);
    out.wt3 = reinterpret_cast<const uint16_t*>(pool.upload(reinterpret_cast<const float*>(planes.data()), planes.size() / 2));
  }
  return out;
}
PackedW pack_upload_ln_1x1(WeightPool& pool, const float* w, const float* bias, const float* lnw, const float* lnb,
                           int M, int K) {
  std::vector<float> wf((size_t)M * K), bf(M);
  for (int o = 0; o < M; ++o) {
    double c = bias ? (double)bias[o] : 0.0;
    for (int k = 0; k < K; ++k) {
      const float wv = w[(size_t)o * K + k];
      wf[(size_t)o * K + k] = wv * lnw[k];
      if (lnb) c += (double)wv * (double)lnb[k];
    }
    bf[o] = (float)c;
  }
  return pack_upload_1x1(pool, wf.data(), (bias || lnb) ? bf.data() : nullptr, M, K);
}
PackedW pack_upload_conv(WeightPool& pool, const float* w, const float* bias, int M, int Cin, int ks) {
  std::vector<float> packed;
  PackedW out;
  const int kp = pack_wt_conv(w, M, Cin, ks, packed, &out.Mpad);
  out.wt = pool.upload(packed);
  out.bias = bias ? pool.upload(bias, M) : nullptr;
  out.M = M; out.K = ks * ks * Cin; out.Cin = Cin;
  if (gemm_split_bf16_enabled() && Cin % 8 == 0 && M >= 17) {   // what the split kernels' conv loaders take
    std::vector<uint16_t> planes;
    out.wt3_plane = (long)pack_split3_kmajor(packed, kp, out.Mpad, planes);
    planes.resize((planes.size() + 1) / 2 * 2, 0);
    out.wt3 = reinterpret_cast<const uint16_t*>(pool.upload(reinterpret_cast<const float*>(planes.data()), planes.size() / 2));
  }
  return out;
}
PackedW pack_upload_3x3(WeightPool& pool, const float* w, const float* bias, int M, int Cin) {
  return pack_upload_conv(pool, w, bias, M, Cin, 3);
}

// ---- kernel clock -----------------------------------------------------------------------------
static thread_local KernelClock* g_clock = nullptr;
KernelClock* current_clock() { return g_clock; }
void set_current_clock(KernelClock* c) { g_clock = c; }

hipEvent_t KernelClock::get_event() {
  if (pool_used == pool.size()) {
    hipEvent_t e;
    JIRF_HIP_CHECK(hipEventCreate(&e));
    pool.push_back(e);
  }
  return pool[pool_used++];
}
void KernelClock::begin_forward() {
  recs.clear();
  pool_used = 0;
}
void KernelClock::collect() {
  agg.clear();
  if (recs.empty()) return;
  JIRF_HIP_CHECK(hipEventSynchronize(recs.back().e1));
  for (auto& r : recs) {
    float ms = 0.f;
    JIRF_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto it = std::find_if(agg.begin(), agg.end(), [&](const Agg& a) { return !std::strcmp(a.name, r.name); });
    if (it == agg.end()) {
      agg.push_back(Agg{r.name, ms, r.flops, r.bytes, 1});
    } else {
      it->ms += ms; it->flops += r.flops; it->bytes += r.bytes; it->launches += 1;
    }
  }
}
KernelClock::~KernelClock() {
  for (auto e : pool) (void)hipEventDestroy(e);
}

const char* intern_label(const char* fmt, ...) {
  static std::mutex mu;
  static std::set<std::string> table;
  char buf[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lock(mu);
  return table.insert(buf).first->c_str();
}

struct ScopeState {
  KernelClock* clock;
  size_t rec;
  hipStream_t s;
};
Scope::Scope(const char* name, double flops, double bytes, hipStream_t s) : h(nullptr) {
  KernelClock* c = current_clock();
  if (!c || !c->enabled) return;
  KernelClock::Rec r{name, flops, bytes, c->get_event(), c->get_event()};
  JIRF_HIP_CHECK(hipEventRecord(r.e0, s));
  c->recs.push_back(r);
  h = new ScopeState{c, c->recs.size() - 1, s};
}
Scope::~Scope() {
  if (!h) return;
  auto* st = static_cast<ScopeState*>(h);
  (void)hipEventRecord(st->clock->recs[st->rec].e1, st->s);
  delete st;
}

// ---- parameters --------------------------------------------------------------------------------
void Net::add_param(const std::string& key, std::vector<int64_t> shape, bool optional) {
  JIRF_REQUIRE(!index.count(key), JIRF_EINVAL, "duplicate parameter key %s", key.c_str());
  ParamSpec p;
  p.key = key;
  p.shape = std::move(shape);
  p.optional = optional;
  index[key] = (int)params.size();
  params.push_back(std::move(p));
}

const std::vector<float>& Net::host(const std::string& key) const {
  auto it = index.find(key);
  JIRF_REQUIRE(it != index.end(), JIRF_ENOTFOUND, "parameter %s is not registered", key.c_str());
  return params[it->second].host;
}

std::string Net::canonical_key(const std::string& key) const {
  // checkpoints saved from a wrapped model carry a "module." prefix
  // (reference basicsr/models/base_model.py:245-249 strips it)
  if (key.rfind("module.", 0) == 0) return key.substr(7);
  return key;
}

int Net::set_param(const char* key_c, const void* data, int dtype, const int64_t* shape, int ndim) {
  JIRF_REQUIRE(key_c && data, JIRF_EINVAL, "set_param: null key or data");
  JIRF_REQUIRE(!finalized, JIRF_ESTATE, "set_param after finalize");
  const std::string key = canonical_key(key_c);
  if (key.empty()) return JIRF_OK;  // accepted and ignored
  auto it = index.find(key);
  JIRF_REQUIRE(it != index.end(), JIRF_ENOTFOUND, "%s has no parameter '%s'", arch.c_str(), key_c);
  ParamSpec& p = params[it->second];
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  bool same = n == p.numel();
  // the reference stores some vectors with trailing singleton axes (e.g. [4,1,1]); require the
  // same element count and the same non-singleton axes
  std::vector<int64_t> a, b;
  for (int i = 0; i < ndim; ++i) if (shape[i] != 1) a.push_back(shape[i]);
  for (auto d : p.shape) if (d != 1) b.push_back(d);
  same = same && a == b;
  if (!same) {
    std::string got, want;
    for (int i = 0; i < ndim; ++i) got += (i ? "," : "") + std::to_string(shape[i]);
    for (size_t i = 0; i < p.shape.size(); ++i) want += (i ? "," : "") + std::to_string(p.shape[i]);
    set_error("parameter '%s': shape [%s] does not match [%s]", key_c, got.c_str(), want.c_str());
    throw Failure{JIRF_EINVAL};
  }
  p.host.resize(n);
  switch (dtype) {
    case JIRF_F32: std::memcpy(p.host.data(), data, n * 4); break;
    case JIRF_F64: for (size_t i = 0; i < n; ++i) p.host[i] = (float)static_cast<const double*>(data)[i]; break;
    case JIRF_I32: for (size_t i = 0; i < n; ++i) p.host[i] = (float)static_cast<const int32_t*>(data)[i]; break;
    case JIRF_I64: for (size_t i = 0; i < n; ++i) p.host[i] = (float)static_cast<const int64_t*>(data)[i]; break;
    default: JIRF_REQUIRE(false, JIRF_EINVAL, "set_param: unknown dtype %d", dtype);
  }
  p.is_set = true;
  return JIRF_OK;
}

void allow_dynamic_lds(DynLdsOnce& once, const void* kernel, int bytes) {
  int dev = 0;
  JIRF_HIP_CHECK(hipGetDevice(&dev));
  const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t)1 << dev : 0;
  if (bit && (once.devices.load(std::memory_order_acquire) & bit)) return;
  JIRF_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (bit) once.devices.fetch_or(bit, std::memory_order_release);
}

}  // namespace jirf
