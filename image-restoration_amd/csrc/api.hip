// extern "C" surface of libjirf_hip.so (declared in include/jirf_hip.h).  Every entry point
// catches jirf::Failure and returns its code; nothing throws across the ABI.
#include <atomic>
#include <cstring>
#include <new>

#include "runtime.h"

using namespace jirf;

struct jirf_net {
  std::unique_ptr<Net> net;
};

namespace {

template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const Failure& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    set_error("host allocation failed");
    return JIRF_ENOMEM;
  } catch (const std::exception& e) {
    set_error("unexpected exception: %s", e.what());
    return JIRF_EINVAL;
  }
}

void require_gfx950(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  JIRF_REQUIRE(e == hipSuccess && n > 0, JIRF_ENOGPU, "no HIP device available (%s)",
               e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  JIRF_REQUIRE(device >= 0 && device < n, JIRF_EINVAL, "device %d out of range (have %d)", device, n);
  hipDeviceProp_t prop;
  JIRF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  JIRF_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, JIRF_ENOGPU,
               "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
  JIRF_HIP_CHECK(hipSetDevice(device));
}

Net* get(jirf_net* n) {
  JIRF_REQUIRE(n && n->net, JIRF_EINVAL, "null network handle");
  return n->net.get();
}
const Net* get(const jirf_net* n) {
  JIRF_REQUIRE(n && n->net, JIRF_EINVAL, "null network handle");
  return n->net.get();
}

}  // namespace

namespace jirf {
// att[b][m][k] = att_t[b][k][m]: the block-diagonal attention of the GEMM weight layout back in natural orientation
__global__ __launch_bounds__(256) void att_t_to_natural_kernel(const float* __restrict__ att_t, float* __restrict__ att,
                                                               int C, int kpad, int mpad) {
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (e >= C * C) return;
  const int m = e / C, k = e - m * C;
  att[(long)b * C * C + e] = att_t[((long)b * kpad + k) * mpad + m];
}
}  // namespace jirf

extern "C" {

int jirf_abi_version(void) { return JIRF_ABI_VERSION; }

int jirf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* jirf_last_error(void) { return last_error(); }

jirf_net* jirf_net_create(const char* arch, const char* cfg_json, int device) {
  jirf_net* out = nullptr;
  const int rc = guarded([&] {
    JIRF_REQUIRE(arch, JIRF_EINVAL, "null arch");
    Config cfg;
    cfg.obj = parse_json_object(cfg_json);
    std::unique_ptr<Net> net;
    if (!std::strcmp(arch, "HINT")) {
      net = make_hint_net(cfg);
    } else if (!std::strcmp(arch, "ASTv2")) {
      net = make_astv2_net(cfg);
    } else if (!std::strcmp(arch, "AST")) {
      net = make_ast_net(cfg);
    } else if (!std::strcmp(arch, "FPro")) {
      net = make_fpro_net(cfg);
    } else {
      // define_network raises ValueError("<type> is not found.") (archs/__init__.py:39-40)
      set_error("%s is not found.", arch);
      throw Failure{JIRF_ENOTFOUND};
    }
    // construction is host-only (so the key layout can be inspected anywhere); the device is
    // checked when the weights are uploaded
    net->device = device;
    out = new jirf_net{std::move(net)};
    return JIRF_OK;
  });
  return rc == JIRF_OK ? out : nullptr;
}

void jirf_net_destroy(jirf_net* net) {
  if (!net) return;
  (void)hipDeviceSynchronize();
  delete net;
}

int jirf_net_set_param(jirf_net* net, const char* key, const void* host_data, int dtype,
                       const int64_t* shape, int ndim) {
  return guarded([&] { return get(net)->set_param(key, host_data, dtype, shape, ndim); });
}

int jirf_net_param_count(const jirf_net* net) {
  int n = -1;
  guarded([&] {
    n = (int)get(net)->params.size();
    return JIRF_OK;
  });
  return n;
}

int jirf_net_param_info(const jirf_net* net, int index, const char** key_out, int64_t* shape_out,
                        int* ndim_out) {
  return guarded([&] {
    const Net* n = get(net);
    JIRF_REQUIRE(index >= 0 && index < (int)n->params.size(), JIRF_EINVAL, "param index %d out of range", index);
    const ParamSpec& p = n->params[index];
    JIRF_REQUIRE(p.shape.size() <= 8, JIRF_EINVAL, "param rank > 8");
    if (key_out) *key_out = p.key.c_str();
    if (ndim_out) *ndim_out = (int)p.shape.size();
    if (shape_out)
      for (size_t i = 0; i < p.shape.size(); ++i) shape_out[i] = p.shape[i];
    return JIRF_OK;
  });
}

int jirf_net_get_param(const jirf_net* net, const char* key, float* host_out, size_t n_floats) {
  return guarded([&] {
    const Net* n = get(net);
    JIRF_REQUIRE(key && host_out, JIRF_EINVAL, "null argument");
    const std::string k = n->canonical_key(key);
    auto it = n->index.find(k);
    JIRF_REQUIRE(it != n->index.end(), JIRF_ENOTFOUND, "%s has no parameter '%s'", n->arch.c_str(), key);
    const ParamSpec& p = n->params[it->second];
    JIRF_REQUIRE(p.is_set, JIRF_ESTATE, "parameter '%s' was never set", key);
    JIRF_REQUIRE(n_floats >= p.host.size(), JIRF_EINVAL, "buffer too small for '%s'", key);
    std::memcpy(host_out, p.host.data(), p.host.size() * 4);
    return JIRF_OK;
  });
}

int jirf_net_finalize(jirf_net* net) {
  return guarded([&] {
    Net* n = get(net);
    JIRF_REQUIRE(!n->finalized, JIRF_ESTATE, "finalize called twice");
    require_gfx950(n->device);
    n->finalize();
    return JIRF_OK;
  });
}

int jirf_net_plan(jirf_net* net, int B, int H, int W, int batch_mode, size_t* workspace_bytes) {
  return guarded([&] {
    Net* n = get(net);
    JIRF_REQUIRE(n->finalized, JIRF_ESTATE, "plan before finalize");
    JIRF_HIP_CHECK(hipSetDevice(n->device));
    const size_t bytes = n->plan(B, H, W, batch_mode);
    if (workspace_bytes) *workspace_bytes = bytes;
    return JIRF_OK;
  });
}

int jirf_net_forward(jirf_net* net, const float* x, float* y, int B, int H, int W, int batch_mode,
                     int on_device, void* stream) {
  return guarded([&] {
    Net* n = get(net);
    JIRF_REQUIRE(x && y, JIRF_EINVAL, "forward: null input or output");
    JIRF_REQUIRE(n->finalized, JIRF_ESTATE, "forward before finalize");
    JIRF_HIP_CHECK(hipSetDevice(n->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (on_device) {
      n->forward(x, y, B, H, W, batch_mode, s);
      if (n->clock.enabled) n->clock.collect();
      return JIRF_OK;
    }
    n->plan(B, H, W, batch_mode);
    const size_t px = (size_t)B * H * W;
    JIRF_HIP_CHECK(hipMemcpyAsync(n->stage_in, x, px * n->io_in_ch * 4, hipMemcpyHostToDevice, s));
    n->forward(n->stage_in, n->stage_out, B, H, W, batch_mode, s);
    JIRF_HIP_CHECK(hipMemcpyAsync(y, n->stage_out, px * n->io_out_ch * 4, hipMemcpyDeviceToHost, s));
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    if (n->clock.enabled) n->clock.collect();
    return JIRF_OK;
  });
}

int jirf_net_reset_cache(jirf_net* net) {
  return guarded([&] {
    get(net)->reset_cache();
    return JIRF_OK;
  });
}
int jirf_net_get_cache(jirf_net* net, float* host_out, int* n_out, int* c_out, size_t capacity_floats) {
  return guarded([&] {
    JIRF_REQUIRE(host_out && n_out && c_out, JIRF_EINVAL, "null argument");
    return get(net)->get_cache(host_out, n_out, c_out, capacity_floats);
  });
}
int jirf_net_set_cache(jirf_net* net, const float* host_in, int n, int c) {
  return guarded([&] { return get(net)->set_cache(host_in, n, c); });
}

int jirf_net_attention_blocks(const jirf_net* net, int* n_blocks, int* channels_out, int capacity) {
  return guarded([&] {
    JIRF_REQUIRE(n_blocks, JIRF_EINVAL, "null argument");
    return const_cast<Net*>(get(net))->attention_blocks(n_blocks, channels_out, capacity);
  });
}
int jirf_net_get_perms(jirf_net* net, int32_t* host_out, size_t capacity_ints, size_t* n_ints) {
  return guarded([&] {
    JIRF_REQUIRE(host_out && n_ints, JIRF_EINVAL, "null argument");
    return get(net)->get_perms(host_out, capacity_ints, n_ints);
  });
}
int jirf_net_set_forced_perms(jirf_net* net, const int32_t* host_in, size_t n_ints) {
  return guarded([&] { return get(net)->set_forced_perms(host_in, n_ints); });
}

int jirf_net_enable_kernel_timing(jirf_net* net, int enable) {
  return guarded([&] {
    get(net)->clock.enabled = enable != 0;
    return JIRF_OK;
  });
}
int jirf_net_kernel_times(jirf_net* net, const char** names_out, float* ms_out, double* flops_out,
                          double* bytes_out, int* launches_out, int capacity, int* n_out) {
  return guarded([&] {
    Net* n = get(net);
    JIRF_REQUIRE(n_out, JIRF_EINVAL, "null argument");
    const auto& agg = n->clock.agg;
    *n_out = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < capacity; ++i) {
      if (names_out) names_out[i] = agg[i].name;
      if (ms_out) ms_out[i] = agg[i].ms;
      if (flops_out) flops_out[i] = agg[i].flops;
      if (bytes_out) bytes_out[i] = agg[i].bytes;
      if (launches_out) launches_out[i] = agg[i].launches;
    }
    return JIRF_OK;
  });
}

// ---- single-kernel entry points -----------------------------------------------------------------

int jirf_op_conv1x1(const float* x, const float* w_host, const float* bias_host, const float* res,
                    float* y, int B, int Cin, int Cout, int P, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && w_host && y && B > 0 && Cin > 0 && Cout > 0 && P > 0, JIRF_EINVAL, "conv1x1: bad arguments");
    WeightPool pool;
    const PackedW w = pack_upload_1x1(pool, w_host, bias_host, Cout, Cin);
    GemmArgs g{};
    g.B = B; g.M = Cout; g.K = Cin; g.P = P; g.Mpad = w.Mpad; g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.bias = w.bias;
    g.x = tv(const_cast<float*>(x), (long)Cin * P);
    g.y = tv(y, (long)Cout * P);
    g.imgs_per_group = 1;
    g.epi = res ? EP_RESID : EP_STORE;
    g.r = tv(const_cast<float*>(res), (long)Cout * P);
    launch_gemm(g, LD_PLAIN, static_cast<hipStream_t>(stream));
    JIRF_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return JIRF_OK;
  });
}

int jirf_op_ln_conv1x1(const float* x, const float* ln_w, const float* ln_b, int ln_kind,
                       const float* w_host, float* y, int B, int Cin, int Cout, int P, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && ln_w && w_host && y && B > 0 && Cin > 0 && Cout > 0 && P > 0, JIRF_EINVAL, "ln_conv1x1: bad arguments");
    JIRF_REQUIRE(ln_kind == 1 || ln_kind == 2, JIRF_EINVAL, "ln_conv1x1: ln_kind %d", ln_kind);
    JIRF_REQUIRE(ln_kind == 2 || ln_b, JIRF_EINVAL, "ln_conv1x1: WithBias needs ln_b");
    hipStream_t s = static_cast<hipStream_t>(stream);
    WeightPool pool;
    // JIRF_OP_LN_FOLD=1 (profiling scripts): the form the networks run - LayerNorm weight and bias folded into the
    // packed weights and the output bias, the kernel only normalises
    static const bool fold = std::getenv("JIRF_OP_LN_FOLD") != nullptr;
    PackedW w;
    if (fold) {
      std::vector<float> hw(Cin), hb(Cin, 0.f);
      JIRF_HIP_CHECK(hipMemcpy(hw.data(), ln_w, (size_t)Cin * 4, hipMemcpyDeviceToHost));
      if (ln_kind == 1) JIRF_HIP_CHECK(hipMemcpy(hb.data(), ln_b, (size_t)Cin * 4, hipMemcpyDeviceToHost));
      w = pack_upload_ln_1x1(pool, w_host, nullptr, hw.data(), ln_kind == 1 ? hb.data() : nullptr, Cout, Cin);
      ln_w = nullptr;
    } else {
      w = pack_upload_1x1(pool, w_host, nullptr, Cout, Cin);
    }
    DeviceBuffer stats;
    stats.ensure((size_t)2 * B * P * 4);
    float* mu = static_cast<float*>(stats.p);
    float* rstd = mu + (size_t)B * P;
    const TV xv = tv(const_cast<float*>(x), (long)Cin * P);
    if (!gemm_ln_inline(Cin, Cout, P, B)) launch_ln_stats(xv, mu, rstd, B, Cin, P, s);
    GemmArgs g{};
    g.B = B; g.M = Cout; g.K = Cin; g.P = P; g.Mpad = w.Mpad; g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane;
    g.x = xv; g.y = tv(y, (long)Cout * P); g.imgs_per_group = 1; g.epi = EP_STORE; g.bias = w.bias;
    g.mu = mu; g.rstd = rstd; g.lnw = ln_w; g.lnb = (ln_w && ln_kind == 1) ? ln_b : nullptr; g.ln_center = ln_kind == 1;
    launch_gemm(g, LD_LN, s);
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_op_gdfn(const float* x, float* y, int B, int C, int hidden, int H, int W, int ln_kind, const float* ln_w_host,
                 const float* ln_b_host, const float* w_in_host, const float* b_in_host, const float* w_dw_host,
                 const float* b_dw_host, const float* w_out_host, const float* b_out_host, int fused, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && y && x != y && B > 0 && C > 0 && hidden > 0 && H > 0 && W > 0 && ln_w_host && w_in_host && w_dw_host &&
                     w_out_host, JIRF_EINVAL, "gdfn: bad arguments");
    JIRF_REQUIRE(ln_kind == 1 || ln_kind == 2, JIRF_EINVAL, "gdfn: ln_kind %d", ln_kind);
    JIRF_REQUIRE(ln_kind == 2 || ln_b_host, JIRF_EINVAL, "gdfn: WithBias needs ln_b");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int P = H * W, h = hidden;
    WeightPool pool;
    const TV xv = tv(const_cast<float*>(x), (long)C * P), yv = tv(y, (long)C * P);
    if (fused) {
      JIRF_REQUIRE(gdfn_fused_supported(C, h, H, W), JIRF_EINVAL,
                   "gdfn: the one-kernel form is built for 96 channels, hidden <= 256 and planes at least 32 wide "
                   "(got C=%d hidden=%d %dx%d)", C, h, H, W);
      const GdfnFusedW w = pack_upload_gdfn_fused(pool, C, h, ln_w_host, ln_kind == 1 ? ln_b_host : nullptr, w_in_host,
                                                  b_in_host, w_dw_host, b_dw_host, w_out_host, b_out_host);
      launch_gdfn_fused(w, xv, yv, B, H, W, ln_kind == 1, s);
    } else {
      // the three launches the networks use for the shapes the one-kernel form does not take
      const PackedW pin = pack_upload_ln_1x1(pool, w_in_host, b_in_host, ln_w_host, ln_kind == 1 ? ln_b_host : nullptr, 2 * h, C);
      const PackedW pout = pack_upload_1x1(pool, w_out_host, b_out_host, C, h);
      const float* dww = pool.upload(w_dw_host, (size_t)2 * h * 9);
      const float* dwb = b_dw_host ? pool.upload(b_dw_host, (size_t)2 * h) : nullptr;
      DeviceBuffer t1, t2, stats;
      t1.ensure((size_t)B * 2 * h * P * 4);
      t2.ensure((size_t)B * h * P * 4);
      stats.ensure((size_t)2 * B * P * 4);
      float* mu = static_cast<float*>(stats.p);
      float* rstd = mu + (size_t)B * P;
      if (!gemm_ln_inline(C)) launch_ln_stats(xv, mu, rstd, B, C, P, s);
      GemmArgs g{};
      g.B = B; g.M = 2 * h; g.K = C; g.P = P; g.Mpad = pin.Mpad; g.wt = pin.wt; g.bias = pin.bias;
      g.x = xv; g.y = tv(static_cast<float*>(t1.p), (long)2 * h * P); g.imgs_per_group = 1; g.epi = EP_STORE;
      g.mu = mu; g.rstd = rstd; g.ln_center = ln_kind == 1;
      launch_gemm(g, LD_LN, s);
      launch_dwconv3x3(g.y, dww, dwb, tv(static_cast<float*>(t2.p), (long)h * P), B, 2 * h, H, W, 1, s);
      GemmArgs o{};
      o.B = B; o.M = C; o.K = h; o.P = P; o.Mpad = pout.Mpad; o.wt = pout.wt; o.bias = pout.bias;
      o.x = tv(static_cast<float*>(t2.p), (long)h * P); o.y = yv; o.imgs_per_group = 1; o.epi = EP_RESID; o.r = xv;
      launch_gemm(o, LD_PLAIN, s);
    }
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

// ---- the entry point that allocates nothing ---------------------------------------------------------
// jirf_op_ln_conv1x1 keeps two things in device memory besides its input and output: the weights in
// the GEMM's K-major packing and, for Cin > 256, the per-pixel LayerNorm statistics.  Here both live in
// one file-scope device array, so the call needs no hipMalloc.  Every GPU the library runs on holds its
// own copy of the array at its own address; that address is looked up once per device.
static __device__ __align__(256) float scratch[JIRF_STATIC_MAX_PACKED_WEIGHT_FLOATS + 2 * JIRF_STATIC_MAX_PIXELS];

namespace {
constexpr size_t STATIC_W_FLOATS = JIRF_STATIC_MAX_PACKED_WEIGHT_FLOATS;
constexpr size_t STATIC_PIXELS = JIRF_STATIC_MAX_PIXELS;

constexpr int STATIC_MAX_DEVICES = 64;
std::atomic<float*> g_scratch_ptr[STATIC_MAX_DEVICES];

// Device address of `scratch` on the current device (the array's name is only a host-side handle).
float* static_scratch() {
  int dev = -1;
  JIRF_HIP_CHECK(hipGetDevice(&dev));
  JIRF_REQUIRE(dev >= 0 && dev < STATIC_MAX_DEVICES, JIRF_EINVAL, "static scratch: device %d out of range", dev);
  float* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* sym = nullptr;
    JIRF_HIP_CHECK(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
    JIRF_REQUIRE(sym != nullptr, JIRF_EHIP, "static scratch: no device address on device %d", dev);
    p = static_cast<float*>(sym);
    g_scratch_ptr[dev].store(p, std::memory_order_release);   // every thread finds the same address
  }
  return p;
}

// w [M][K] (reference layout, device) -> wt [Kp][Mp] K-major, zero padded: every element of the packed
// image is written, so nothing of an earlier, larger call survives in the padding.
__global__ __launch_bounds__(256) void pack_wt_1x1_kernel(const float* __restrict__ w, float* __restrict__ wt, int M,
                                                          int K, int Mp, int Kp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Kp * Mp) return;
  const int k = i / Mp, o = i - k * Mp;
  wt[i] = (k < K && o < M) ? w[(long)o * K + k] : 0.f;
}
}  // namespace

int jirf_op_ln_conv1x1_static(const float* x, const float* ln_w, const float* ln_b, int ln_kind,
                              const float* w_dev, float* y, int B, int Cin, int Cout, int P, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && ln_w && w_dev && y && B > 0 && Cin > 0 && Cout > 0 && P > 0, JIRF_EINVAL,
                 "ln_conv1x1_static: bad arguments");
    JIRF_REQUIRE(ln_kind == 1 || ln_kind == 2, JIRF_EINVAL, "ln_conv1x1_static: ln_kind %d", ln_kind);
    JIRF_REQUIRE(ln_kind == 2 || ln_b, JIRF_EINVAL, "ln_conv1x1_static: WithBias needs ln_b");
    const long Kp = (Cin + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
    const long Mp = (Cout + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
    JIRF_REQUIRE((size_t)(Kp * Mp) <= STATIC_W_FLOATS, JIRF_EINVAL,
                 "ln_conv1x1_static: packed weights of Cin=%d Cout=%d take %ld floats, the static scratch holds %zu "
                 "(use jirf_op_ln_conv1x1)", Cin, Cout, Kp * Mp, STATIC_W_FLOATS);
    JIRF_REQUIRE((size_t)B * (size_t)P <= STATIC_PIXELS, JIRF_EINVAL,
                 "ln_conv1x1_static: B*P = %zu pixels, the static scratch holds statistics for %zu "
                 "(use jirf_op_ln_conv1x1)", (size_t)B * (size_t)P, STATIC_PIXELS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_scratch();
    float* wt = base;
    float* mu = base + STATIC_W_FLOATS;
    float* rstd = mu + (size_t)B * P;
    hipLaunchKernelGGL(pack_wt_1x1_kernel, dim3((unsigned)((Kp * Mp + 255) / 256)), dim3(256), 0, s, w_dev, wt, Cout, Cin,
                       (int)Mp, (int)Kp);
    JIRF_HIP_CHECK(hipGetLastError());
    const TV xv = tv(const_cast<float*>(x), (long)Cin * P);
    if (!gemm_ln_inline(Cin)) launch_ln_stats(xv, mu, rstd, B, Cin, P, s);
    GemmArgs g{};
    g.B = B; g.M = Cout; g.K = Cin; g.P = P; g.Mpad = (int)Mp; g.wt = wt;
    g.x = xv; g.y = tv(y, (long)Cout * P); g.imgs_per_group = 1; g.epi = EP_STORE;
    g.mu = mu; g.rstd = rstd; g.lnw = ln_w; g.lnb = ln_kind == 1 ? ln_b : nullptr; g.ln_center = ln_kind == 1;
    launch_gemm(g, LD_LN, s);
    return JIRF_OK;
  });
}

int jirf_op_conv3x3(const float* x, const float* w_host, const float* bias_host, const float* res,
                    float* y, int B, int Cin, int Cout, int H, int W, int mode, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && w_host && y && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, JIRF_EINVAL, "conv3x3: bad arguments");
    JIRF_REQUIRE(mode >= 0 && mode <= 2, JIRF_EINVAL, "conv3x3: mode %d", mode);
    JIRF_REQUIRE(mode != 1 || (H % 2 == 0 && W % 2 == 0), JIRF_EINVAL, "conv3x3: unshuffle needs even H, W");
    JIRF_REQUIRE(mode != 2 || Cout % 4 == 0, JIRF_EINVAL, "conv3x3: shuffle needs Cout %% 4 == 0");
    WeightPool pool;
    const PackedW w = pack_upload_3x3(pool, w_host, bias_host, Cout, Cin);
    const long P = (long)H * W;
    GemmArgs g{};
    g.B = B; g.M = Cout; g.K = 9 * Cin; g.P = (int)P; g.Mpad = w.Mpad; g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane;
    g.bias = w.bias;
    g.x = tv(const_cast<float*>(x), (long)Cin * P);
    g.y = tv(y, (long)Cout * P);  // same element count in all three modes
    g.H = H; g.W = W; g.Cin = Cin; g.imgs_per_group = 1;
    g.epi = mode == 1 ? EP_UNSHUFFLE : mode == 2 ? EP_SHUFFLE : (res ? EP_RESID : EP_STORE);
    g.r = tv(const_cast<float*>(res), (long)Cout * P);
    launch_gemm(g, LD_CONV3, static_cast<hipStream_t>(stream));
    JIRF_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return JIRF_OK;
  });
}

int jirf_op_gelu_error_scan(float hi, double* out4_host, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(out4_host, JIRF_EINVAL, "gelu_error_scan: null output");
    gelu_error_scan(hi, out4_host, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_dwconv3x3(const float* x, const float* w, const float* bias, float* y, int B, int C,
                      int H, int W, int gate, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && w && y && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "dwconv3x3: bad arguments");
    const long P = (long)H * W;
    launch_dwconv3x3(tv(const_cast<float*>(x), (long)C * P), w, bias, tv(y, (long)(gate ? C / 2 : C) * P), B, C, H, W,
                     gate, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_layernorm(const float* x, const float* w, const float* b, float* y, int B, int C, int P,
                      int kind, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && w && y && B > 0 && C > 0 && P > 0, JIRF_EINVAL, "layernorm: bad arguments");
    launch_layernorm(tv(const_cast<float*>(x), (long)C * P), w, b, tv(y, (long)C * P), B, C, P, kind,
                     static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

// ---- script-level plumbing -------------------------------------------------------------------------
namespace {
TileGrid make_grid(int crop, const int32_t* hstarts, int nh, const int32_t* wstarts, int nw) {
  JIRF_REQUIRE(hstarts && wstarts && nh >= 1 && nw >= 1 && nh <= TILE_GRID_MAX && nw <= TILE_GRID_MAX, JIRF_EINVAL,
               "tile grid: %d x %d starts (at most %d per axis)", nh, nw, TILE_GRID_MAX);
  TileGrid g{};
  g.crop = crop; g.nh = nh; g.nw = nw;
  for (int i = 0; i < nh; ++i) g.hs[i] = hstarts[i];
  for (int i = 0; i < nw; ++i) g.ws[i] = wstarts[i];
  return g;
}
}  // namespace

int jirf_op_reflect_pad(const float* x, float* y, int B, int C, int H, int W, int pad_bottom, int pad_right,
                        void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "reflect_pad: bad arguments");
    launch_reflect_pad(x, y, B, C, H, W, pad_bottom, pad_right, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_split_tiles(const float* x, float* tiles, int B, int C, int H, int W, int crop, const int32_t* hstarts,
                        int nh, const int32_t* wstarts, int nw, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && tiles && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "split_tiles: bad arguments");
    launch_split_tiles(x, tiles, make_grid(crop, hstarts, nh, wstarts, nw), B, C, H, W,
                       static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_merge_tiles(const float* tiles, float* out, int B, int C, int H, int W, int crop, const int32_t* hstarts,
                        int nh, const int32_t* wstarts, int nw, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(tiles && out && B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "merge_tiles: bad arguments");
    launch_merge_tiles(tiles, out, make_grid(crop, hstarts, nh, wstarts, nw), B, C, H, W,
                       static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_crop_clamp(const float* x, float* y, int B, int C, int H, int W, int h, int w, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && y && B > 0 && C > 0, JIRF_EINVAL, "crop_clamp: bad arguments");
    launch_crop_clamp(x, y, B, C, H, W, h, w, false, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_psnr_work_doubles(void) { return (int)psnr_work_doubles(); }

int jirf_op_psnr_stats(const float* a, const float* b, int C, int H, int W, int crop_border, int quantise, int y_channel,
                       int blue_plane, double* work, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(a && b && work && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "psnr_stats: bad arguments");
    launch_psnr_stats(a, b, C, H, W, crop_border, quantise, y_channel, blue_plane, work, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_ssim_work_doubles(int C, int H, int W) { return (int)ssim_work_doubles(C, H, W); }

int jirf_op_ssim_stats(const float* a, const float* b, int C, int H, int W, int crop_border, int quantise, int y_channel,
                       int blue_plane, float max_value, double* work, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(a && b && work && C > 0 && H > 0 && W > 0 && max_value > 0.f, JIRF_EINVAL, "ssim_stats: bad arguments");
    launch_ssim_stats(a, b, C, H, W, crop_border, quantise, y_channel, blue_plane, max_value, work,
                      static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_crop_clamp_u8(const float* x, uint8_t* y, int B, int C, int H, int W, int h, int w, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && y && B > 0 && C > 0, JIRF_EINVAL, "crop_clamp_u8: bad arguments");
    launch_crop_clamp(x, y, B, C, H, W, h, w, true, static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_hint_attention_core(const float* qkv, const float* params_host, const float* cache_in,
                                int cache_c, float* out, float* tcache, int32_t* perm, float* att,
                                float* cache_out, int B, int C, int P, int batch_mode,
                                void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(qkv && params_host && out && tcache && perm && att && cache_out, JIRF_EINVAL,
                 "hint_attention_core: null argument");
    JIRF_REQUIRE(B > 0 && C >= 8 && P >= 2, JIRF_EINVAL, "hint_attention_core: bad shape");
    JIRF_REQUIRE(!cache_in || cache_c > 0, JIRF_EINVAL, "hint_attention_core: cache size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const HintAttnShape sh = hint_attn_shape(B, C, P, batch_mode);
    WeightPool pool;
    HintAttnWeights w{};
    const float* p = params_host;
    w.temperature = pool.upload(p, 4);
    p += 4;
    for (int g = 0; g < 4; ++g) {
      const int c = sh.gsz[g];
      std::vector<float> packed;
      pack_hint_modulator(p, 2 * c, c, packed);
      w.cw_wt[g] = pool.upload(packed); p += (size_t)2 * c * c;
      w.cw_b[g] = pool.upload(p, (size_t)2 * c); p += 2 * c;
      pack_hint_modulator(p, c, c, packed);
      w.gc_wt[g] = pool.upload(packed); p += (size_t)c * c;
      w.gc_b[g] = pool.upload(p, (size_t)c); p += c;
    }
    HintAttnScratch sc;
    DeviceBuffer scratch;
    scratch.ensure(hint_attn_scratch_carve(sh, nullptr, &sc));
    hint_attn_scratch_carve(sh, scratch.p, &sc);
    const TV t = tv(const_cast<float*>(qkv), 3L * C * P);
    launch_hint_attention(sh, tv_chan(t, 0, P), tv_chan(t, C, P), w, cache_in, cache_c, cache_out, sc,
                          perm, nullptr, s);
    GemmArgs g{};
    g.B = B; g.M = C; g.K = C; g.P = P;
    g.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
    g.wt = sc.att_t;
    g.wt_bs = (long)((C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD) * g.Mpad;
    g.x = tv_chan(t, 2 * C, P);
    g.kidx = perm; g.imgs_per_group = B / sh.G;
    g.y = tv(out, (long)C * P); g.y2 = tv(tcache, (long)C * P);
    g.epi = EP_AV_INTRA; g.q = tv_chan(t, 0, P); g.k = tv_chan(t, C, P);
    g.rq = sc.rq; g.rk = sc.rk;
    launch_gemm(g, LD_GATHER, s);
    JIRF_HIP_CHECK(hipMemcpyAsync(att, sc.att_nat, (size_t)B * C * C * 4, hipMemcpyDeviceToDevice, s));
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_op_head_attention_core(const float* qkv, const float* temperature_host, const float* ca_w_host,
                                const float* ca_b_host, float norm_add_eps, float* out, float* att, int B, int C,
                                int P, int heads, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(qkv && temperature_host && out && att, JIRF_EINVAL, "head_attention_core: null argument");
    JIRF_REQUIRE(B > 0 && C > 0 && P >= 2 && heads > 0 && C % heads == 0, JIRF_EINVAL, "head_attention_core: bad shape");
    JIRF_REQUIRE(ca_w_host || !ca_b_host, JIRF_EINVAL, "head_attention_core: att_ca bias without weight");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ch = C / heads;
    WeightPool pool;
    const float* temperature = pool.upload(temperature_host, (size_t)heads);
    const float* ca_w = ca_w_host ? pool.upload(ca_w_host, (size_t)2 * ch * ch) : nullptr;
    const float* ca_b = ca_b_host ? pool.upload(ca_b_host, (size_t)2 * ch) : nullptr;
    HeadAttnScratch sc;
    const int splits = head_attn_splits(B, C, P);
    DeviceBuffer scratch;
    scratch.ensure(head_attn_scratch_carve(B, C, P, splits, nullptr, &sc));
    head_attn_scratch_carve(B, C, P, splits, scratch.p, &sc);
    const TV t = tv(const_cast<float*>(qkv), 3L * C * P);
    launch_head_attention(B, C, P, heads, tv_chan(t, 0, P), tv_chan(t, C, P), temperature, ca_w, ca_b, sc, s,
                          norm_add_eps);
    GemmArgs g{};
    g.B = B; g.M = C; g.K = C; g.P = P;
    g.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
    const int kpad = (C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD;
    g.wt = sc.att_t;
    g.wt_bs = (long)kpad * g.Mpad;
    g.x = tv_chan(t, 2 * C, P); g.y = tv(out, (long)C * P);
    g.epi = EP_STORE; g.imgs_per_group = 1;
    launch_gemm(g, LD_PLAIN, s);
    // the attention in natural orientation: att[b][m][k] = att_t[b][k][m]
    hipLaunchKernelGGL(att_t_to_natural_kernel, dim3((unsigned)((C * C + 255) / 256), (unsigned)B), dim3(256), 0, s,
                       sc.att_t, att, C, kpad, g.Mpad);
    JIRF_HIP_CHECK(hipGetLastError());
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_op_cross_window_attention(const float* q, const float* kv, float* out, int B, int C, int heads, int H,
                                   int W, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(q && kv && out, JIRF_EINVAL, "cross_window_attention: null argument");
    JIRF_REQUIRE(B > 0 && C > 0 && heads > 0, JIRF_EINVAL, "cross_window_attention: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long P = (long)H * W;
    launch_cross_window_attention(tv(const_cast<float*>(q), C * P), tv(const_cast<float*>(kv), 2L * C * P),
                                  tv(out, C * P), B, C, heads, H, W, s);
    return JIRF_OK;
  });
}

int jirf_op_dynamic_filter(const float* x, const float* params_host, int groups, float* low, float* high, int B,
                           int C, int H, int W, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && params_host && low && high, JIRF_EINVAL, "dynamic_filter: null argument");
    JIRF_REQUIRE(B > 0 && groups > 0 && C % groups == 0 && H >= 2 && W >= 2, JIRF_EINVAL, "dynamic_filter: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t F = (size_t)9 * groups;
    WeightPool pool;
    DynFilterW w{};
    const float* p = params_host;
    w.conv = pool.upload(p, F * C); p += F * C;
    w.gate = pool.upload(p, F * F); p += F * F;
    w.bn_w = pool.upload(p, F); p += F;
    w.bn_b = pool.upload(p, F); p += F;
    w.bn_mean = pool.upload(p, F); p += F;
    w.bn_var = pool.upload(p, F);
    w.groups = groups;
    DeviceBuffer scratch;
    scratch.ensure(((size_t)B * C + (size_t)B * F) * sizeof(float));
    float* pooled = static_cast<float*>(scratch.p);
    const long P = (long)H * W;
    launch_dynfilter(tv(const_cast<float*>(x), C * P), w, pooled, pooled + (size_t)B * C, tv(low, C * P), tv(high, C * P),
                     B, C, H, W, s);
    JIRF_HIP_CHECK(hipStreamSynchronize(s));
    return JIRF_OK;
  });
}

int jirf_op_resize(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo, int bicubic, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(x && y, JIRF_EINVAL, "resize: null argument");
    JIRF_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, JIRF_EINVAL, "resize: bad shape");
    launch_resize(x, (long)C * Hi * Wi, (long)Hi * Wi, y, (long)C * Ho * Wo, B, C, Hi, Wi, Ho, Wo, bicubic != 0,
                  static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

int jirf_op_fspg_high(const float* f, const float* k1, const float* b1, const float* k2, const float* b2,
                      const float* wmap, float* y, int B, int C, int H, int W, void* stream) {
  return guarded([&] {
    JIRF_REQUIRE(f && k1 && b1 && k2 && b2 && wmap && y, JIRF_EINVAL, "fspg_high: null argument");
    JIRF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, JIRF_EINVAL, "fspg_high: bad shape");
    const long P = (long)H * W;
    launch_fspg_high(tv(const_cast<float*>(f), C * P), k1, b1, k2, b2, wmap, tv(y, C * P), B, C, H, W,
                     static_cast<hipStream_t>(stream));
    return JIRF_OK;
  });
}

}  // extern "C"
