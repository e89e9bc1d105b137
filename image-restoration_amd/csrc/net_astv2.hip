// ASTv2 forward graph (reference basicsr/models/archs/astv2_arch.py:302-469) assembled from the
// libjirf_hip kernels.  Host code only.  Three resolution levels; encoder blocks are FFN-only,
// decoder_level3 runs at the encoder_level3 resolution, in place on its output.
//
// Buffer plan (per image; P1 = H*W, P2 = P1/4, P3 = P1/16, d = dim):
//   cat1 [2d][P1]  upper half = encoder level 1 (in place), lower half = up2_1 output
//   cat2 [4d][P2]  upper half = encoder level 2, lower half = up3_2 output
//   x2 [2d][P2]    reduce_chan_level2 output (decoder level 2 in place)
//   x3 [4d][P3]    encoder level 3 then decoder level 3, in place
//   t1, t2         block temporaries
#include <algorithm>

#include "runtime.h"

namespace jirf {
namespace {

struct Ast2Block {
  std::string name;
  int C = 0, hidden = 0, heads = 1;
  bool att = false;
  const float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
  PackedW qkv, proj, pin, pout;
  const float *qkv_dw_w = nullptr, *qkv_dw_b = nullptr, *ffn_dw_w = nullptr, *ffn_dw_b = nullptr;
  const float *temperature = nullptr, *ca_w = nullptr, *ca_b = nullptr, *partial_w = nullptr;
};
struct Ast2Stage {
  std::string name;
  int level;
  std::vector<Ast2Block> blocks;
};

class Astv2Net final : public Net {
 public:
  explicit Astv2Net(const Config& c) {
    arch = "ASTv2";
    cfg = c;
    static const char* known[] = {"inp_channels", "out_channels", "dim", "num_blocks", "num_refinement_blocks",
                                  "heads", "ffn_expansion_factor", "bias", "ln_type", "dual_pixel_task"};
    for (auto& kv : cfg.obj) {
      bool ok = false;
      for (auto k : known) ok = ok || kv.first == k;
      JIRF_REQUIRE(ok, JIRF_EINVAL, "ASTv2 got an unexpected keyword argument '%s'", kv.first.c_str());
    }
    inp_ch_ = cfg.get_int("inp_channels", 3);
    out_ch_ = cfg.get_int("out_channels", 3);
    dim_ = cfg.get_int("dim", 48);
    nb_ = cfg.get_ints("num_blocks", {4, 6, 6, 8});
    refine_ = cfg.get_int("num_refinement_blocks", 4);
    heads_ = cfg.get_ints("heads", {1, 2, 4, 8});
    ffn_ = cfg.get_num("ffn_expansion_factor", 2.66);
    bias_ = cfg.get_bool("bias", false);
    with_bias_ln_ = cfg.get_str("ln_type", "WithBias") != "BiasFree";
    dual_pixel_ = cfg.get_bool("dual_pixel_task", false);
    JIRF_REQUIRE(nb_.size() >= 3 && heads_.size() >= 3, JIRF_EINVAL, "ASTv2: num_blocks and heads need 3 entries");
    // FeedForward views the tensor as 12 groups and convolves half of each (astv2_arch.py:82-84)
    JIRF_REQUIRE(dim_ > 0 && dim_ % 24 == 0, JIRF_EINVAL, "ASTv2: dim=%d must be a positive multiple of 24", dim_);
    JIRF_REQUIRE(dual_pixel_ || inp_ch_ == out_ch_, JIRF_EINVAL,
                 "ASTv2: the output residual needs inp_channels == out_channels");
    register_params();
  }

  void finalize() override {
    for (auto& p : params)
      JIRF_REQUIRE(p.is_set || p.optional, JIRF_ESTATE, "finalize: parameter '%s' was never set", p.key.c_str());
    auto conv1 = [&](const std::string& k, int M, int K) {
      return pack_upload_1x1(pool_, host(k + ".weight").data(), bias_ ? host(k + ".bias").data() : nullptr, M, K);
    };
    auto conv1_ln = [&](const std::string& k, const std::string& ln, int M, int K) {  // LayerNorm affine folded in
      return pack_upload_ln_1x1(pool_, host(k + ".weight").data(), bias_ ? host(k + ".bias").data() : nullptr,
                                host(ln + ".body.weight").data(),
                                with_bias_ln_ ? host(ln + ".body.bias").data() : nullptr, M, K);
    };
    auto conv3 = [&](const std::string& k, int M, int Cin, bool b) {
      return pack_upload_3x3(pool_, host(k + ".weight").data(), b ? host(k + ".bias").data() : nullptr, M, Cin);
    };
    auto vec = [&](const std::string& k) { return pool_.upload(host(k)); };
    for (auto& st : stages_)
      for (auto& bl : st.blocks) {
        const std::string& n = bl.name;
        const int C = bl.C, h = bl.hidden;
        if (bl.att) {
          bl.n1w = vec(n + ".norm1.body.weight");
          if (with_bias_ln_) bl.n1b = vec(n + ".norm1.body.bias");
          const std::string a = n + ".attn";
          bl.qkv = conv1_ln(a + ".qkv", n + ".norm1", 3 * C, C);
          bl.qkv_dw_w = vec(a + ".qkv_dwconv.weight");
          if (bias_) bl.qkv_dw_b = vec(a + ".qkv_dwconv.bias");
          bl.proj = conv1(a + ".project_out", C, C);
          bl.temperature = vec(a + ".temperature");
          bl.ca_w = vec(a + ".att_ca.weight");
          if (bias_) bl.ca_b = vec(a + ".att_ca.bias");
        }
        bl.n2w = vec(n + ".norm2.body.weight");
        if (with_bias_ln_) bl.n2b = vec(n + ".norm2.body.bias");
        bl.partial_w = vec(n + ".ffn.partial_conv3.weight");
        bl.pin = conv1(n + ".ffn.project_in", 2 * h, C);
        bl.ffn_dw_w = vec(n + ".ffn.dwconv.weight");
        if (bias_) bl.ffn_dw_b = vec(n + ".ffn.dwconv.bias");
        bl.pout = conv1(n + ".ffn.project_out", C, h);
      }
    const int d = dim_;
    patch_ = conv3("patch_embed.proj", d, inp_ch_, false);
    down12_ = conv3("down1_2.body.0", d / 2, d, false);
    down23_ = conv3("down2_3.body.0", d, 2 * d, false);
    up32_ = conv3("up3_2.body.0", 8 * d, 4 * d, false);
    up21_ = conv3("up2_1.body.0", 4 * d, 2 * d, false);
    red2_ = conv1("reduce_chan_level2", 2 * d, 4 * d);
    if (dual_pixel_) skip_ = conv1("skip_conv", 2 * d, d);
    out_ = conv3("output", out_ch_, 2 * d, bias_);
    finalized = true;
  }

  size_t plan(int B, int H, int W, int mode) override {
    JIRF_REQUIRE(finalized, JIRF_ESTATE, "plan before finalize");
    JIRF_REQUIRE(B > 0 && H > 0 && W > 0, JIRF_EINVAL, "plan: bad shape B=%d H=%d W=%d", B, H, W);
    JIRF_REQUIRE(H % 4 == 0 && W % 4 == 0, JIRF_EINVAL,
                 "ASTv2 needs H and W to be multiples of 4, got %dx%d (pad the input as the test scripts do)", H, W);
    (void)mode;  // ASTv2 has no cross-sample coupling: both batch modes are the same computation
    if (B == pB_ && H == pH_ && W == pW_) return arena_.bytes;
    Arena m;
    carve(m, B, H, W);
    arena_.ensure(m.off);
    Arena a;
    a.base = static_cast<unsigned char*>(arena_.p);
    carve(a, B, H, W);
    pB_ = B; pH_ = H; pW_ = W;
    stage_in = in_; stage_out = outb_; io_in_ch = inp_ch_; io_out_ch = out_ch_;
    return arena_.bytes;
  }

  void forward(const float* x, float* y, int B, int H, int W, int mode, hipStream_t s) override {
    plan(B, H, W, mode);
    set_current_clock(&clock);
    clock.begin_forward();
    try {
      run(x, y, B, H, W, s);
    } catch (...) {
      set_current_clock(nullptr);
      throw;
    }
    set_current_clock(nullptr);
  }

 private:
  void register_params() {
    const int d = dim_;
    add_param("patch_embed.proj.weight", {d, inp_ch_, 3, 3});
    struct SP { const char* name; int n; int mult; int hi; bool att; int level; };
    const SP plan[] = {{"encoder_level1", nb_[0], 1, 0, false, 1}, {"encoder_level2", nb_[1], 2, 1, false, 2},
                       {"encoder_level3", nb_[2], 4, 2, false, 3}, {"decoder_level3", nb_[2], 4, 2, true, 3},
                       {"decoder_level2", nb_[1], 2, 1, true, 2},  {"decoder_level1", nb_[0], 2, 0, true, 1},
                       {"refinement", refine_, 2, 0, true, 1}};
    for (auto& sp : plan) {
      Ast2Stage st;
      st.name = sp.name;
      st.level = sp.level;
      for (int i = 0; i < sp.n; ++i) {
        Ast2Block bl;
        bl.name = std::string(sp.name) + "." + std::to_string(i);
        bl.C = d * sp.mult;
        bl.hidden = (int)((double)bl.C * ffn_);
        bl.heads = heads_[sp.hi];
        bl.att = sp.att;
        JIRF_REQUIRE(!bl.att || bl.C % bl.heads == 0, JIRF_EINVAL, "ASTv2: %d channels are not divisible by %d heads",
                     bl.C, bl.heads);
        register_block(bl);
        st.blocks.push_back(std::move(bl));
      }
      stages_.push_back(std::move(st));
    }
    auto conv = [&](const std::string& k, int co, int ci, int ks, bool b) {
      add_param(k + ".weight", {co, ci, ks, ks});
      if (b) add_param(k + ".bias", {co});
    };
    conv("down1_2.body.0", d / 2, d, 3, false);
    conv("down2_3.body.0", d, 2 * d, 3, false);
    conv("up3_2.body.0", 8 * d, 4 * d, 3, false);
    conv("up2_1.body.0", 4 * d, 2 * d, 3, false);
    conv("reduce_chan_level2", 2 * d, 4 * d, 1, bias_);
    if (dual_pixel_) conv("skip_conv", 2 * d, d, 1, bias_);
    conv("output", out_ch_, 2 * d, 3, bias_);
  }

  void register_block(const Ast2Block& bl) {
    const std::string& n = bl.name;
    const int C = bl.C, h = bl.hidden;
    auto ln = [&](const std::string& k) {
      add_param(k + ".body.weight", {C});
      if (with_bias_ln_) add_param(k + ".body.bias", {C});
    };
    auto conv = [&](const std::string& k, int co, int ci, int ks) {
      add_param(k + ".weight", {co, ci, ks, ks});
      if (bias_) add_param(k + ".bias", {co});
    };
    if (bl.att) {
      ln(n + ".norm1");
      const std::string a = n + ".attn";
      const int ch = C / bl.heads;
      add_param(a + ".temperature", {bl.heads, 1, 1});
      conv(a + ".qkv", 3 * C, C, 1);
      conv(a + ".qkv_dwconv", 3 * C, 1, 3);
      conv(a + ".project_out", C, C, 1);
      conv(a + ".att_ca", 2 * ch, ch, 1);
    }
    ln(n + ".norm2");
    add_param(n + ".ffn.partial_conv3.weight", {C / 24, 1, 3, 3});
    conv(n + ".ffn.project_in", 2 * h, C, 1);
    conv(n + ".ffn.dwconv", h, 1, 3);
    conv(n + ".ffn.project_out", C, h, 1);
  }

  void carve(Arena& a, int B, int H, int W) {
    splitk_ = a.floats(SPLITK_WS_FLOATS);   // slabs of split-K GEMMs (launch_gemm)
    const size_t d = dim_, P1 = (size_t)H * W, P2 = P1 / 4, P3 = P1 / 16, nB = B;
    in_ = a.floats(nB * inp_ch_ * P1);
    outb_ = a.floats(nB * out_ch_ * P1);
    cat1_ = a.floats(nB * 2 * d * P1);
    cat2_ = a.floats(nB * 4 * d * P2);
    x2_ = a.floats(nB * 2 * d * P2);
    x3_ = a.floats(nB * 4 * d * P3);
    emb_ = dual_pixel_ ? a.floats(nB * d * P1) : nullptr;
    size_t t1 = 0, t2 = 0, scratch = 0;
    for (auto& st : stages_) {
      const size_t P = st.level == 1 ? P1 : st.level == 2 ? P2 : P3;
      for (auto& bl : st.blocks) {
        t1 = std::max(t1, (size_t)std::max(3 * bl.C, 2 * bl.hidden) * P);
        t2 = std::max(t2, (size_t)std::max(3 * bl.C, bl.hidden) * P);
        if (bl.att) {
          HeadAttnScratch tmp;
          scratch = std::max(scratch, head_attn_scratch_carve(B, bl.C, (int)P, head_attn_splits(B, bl.C, (int)P), nullptr, &tmp));
        }
      }
    }
    t1_ = a.floats(nB * t1);
    t2_ = a.floats(nB * t2);
    t1_per_ = t1; t2_per_ = t2;
    mu_ = a.floats(nB * P1);
    rstd_ = a.floats(nB * P1);
    attn_scratch_ = a.take(scratch);
  }

  void gemm(const PackedW& w, TV x, TV y, int B, int P, int epi, TV r, hipStream_t s, int loader = LD_PLAIN,
            const float* lnw = nullptr, const float* lnb = nullptr) {
    GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
    g.B = B; g.M = w.M; g.K = w.K; g.P = P; g.Mpad = w.Mpad;
    g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.bias = w.bias; g.x = x; g.y = y; g.epi = epi; g.r = r; g.imgs_per_group = 1;
    if (loader == LD_LN) {
      g.mu = mu_; g.rstd = rstd_; g.lnw = lnw; g.lnb = lnb; g.ln_center = with_bias_ln_ ? 1 : 0;
    }
    launch_gemm(g, loader, s);
  }
  void conv3(const PackedW& w, TV x, TV y, int B, int H, int W, int epi, TV r, hipStream_t s) {
    GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
    g.B = B; g.M = w.M; g.K = w.K; g.P = H * W; g.Mpad = w.Mpad;
    g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.bias = w.bias; g.x = x; g.y = y; g.epi = epi; g.r = r;
    g.H = H; g.W = W; g.Cin = w.Cin; g.imgs_per_group = 1;
    launch_gemm(g, LD_CONV3, s);
  }

  void run_block(const Ast2Block& bl, TV x, int B, int H, int W, hipStream_t s) {
    const int C = bl.C, P = H * W, h = bl.hidden;
    const TV t1 = tv(t1_, (long)t1_per_), t2 = tv(t2_, (long)t2_per_);
    if (bl.att) {
      if (!gemm_ln_inline(C, bl.qkv.M, P, B)) launch_ln_stats(x, mu_, rstd_, B, C, P, s);
      gemm(bl.qkv, x, t1, B, P, EP_STORE, TV{}, s, LD_LN);   // norm1 folded into bl.qkv
      launch_dwconv3x3(t1, bl.qkv_dw_w, bl.qkv_dw_b, t2, B, 3 * C, H, W, 0, s);
      HeadAttnScratch sc;
      head_attn_scratch_carve(B, C, P, head_attn_splits(B, C, P), attn_scratch_, &sc);
      launch_head_attention(B, C, P, bl.heads, tv_chan(t2, 0, P), tv_chan(t2, C, P), bl.temperature, bl.ca_w,
                            bl.ca_b, sc, s);
      GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;  // attention * V: block-diagonal per-image "weight"
      g.B = B; g.M = C; g.K = C; g.P = P;
      g.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
      g.wt = sc.att_t;
      g.wt_bs = (long)((C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD) * g.Mpad;
      g.x = tv_chan(t2, 2 * C, P); g.y = tv_chan(t1, 0, P);
      g.epi = EP_STORE; g.imgs_per_group = 1;
      if (gemm_split_bf16_enabled() && C <= 96 && (long)((P + 127) / 128) * B >= 512) {
        // the split kernel for short outputs takes per-image weights as per-image planes, made here on the device
        launch_split_planes(sc.att_t, sc.att3, B, (C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD, g.Mpad, s);
        g.wt3 = sc.att3; g.wt3_plane = g.wt_bs; g.wt3_bs = 3 * g.wt_bs;
      }
      launch_gemm(g, LD_PLAIN, s);
      gemm(bl.proj, tv_chan(t1, 0, P), x, B, P, EP_RESID, x, s);
    }
    // FeedForward: LN -> partial conv -> 1x1 up -> dw3x3(first half) * second half -> 1x1 down
    launch_layernorm(x, bl.n2w, bl.n2b, t1, B, C, P, with_bias_ln_ ? 1 : 2, s);
    launch_dwconv3x3(t1, bl.partial_w, nullptr, t2, B, C, H, W, 0, s, C / 12, C / 24);
    gemm(bl.pin, t2, t1, B, P, EP_STORE, TV{}, s);
    launch_dwconv3x3(t1, bl.ffn_dw_w, bl.ffn_dw_b, t2, B, 2 * h, H, W, 2, s);
    gemm(bl.pout, t2, x, B, P, EP_RESID, x, s);
  }

  void run(const float* x, float* y, int B, int H, int W, hipStream_t s) {
    const int d = dim_;
    const long P1 = (long)H * W, P2 = P1 / 4, P3 = P1 / 16;
    const int H2 = H / 2, W2 = W / 2, H3 = H / 4, W3 = W / 4;
    const TV xin = tv(const_cast<float*>(x), (long)inp_ch_ * P1);
    const TV yout = tv(y, (long)out_ch_ * P1);
    const TV cat1 = tv(cat1_, 2L * d * P1), cat2 = tv(cat2_, 4L * d * P2);
    const TV x2 = tv(x2_, 2L * d * P2), x3 = tv(x3_, 4L * d * P3);
    const TV e1 = tv_chan(cat1, d, P1), e2 = tv_chan(cat2, 2 * d, P2);
    auto run_stage = [&](int idx, TV t, int h, int w) {
      for (auto& bl : stages_[idx].blocks) run_block(bl, t, B, h, w, s);
    };
    conv3(patch_, xin, e1, B, H, W, EP_STORE, TV{}, s);
    if (dual_pixel_)
      JIRF_HIP_CHECK(hipMemcpy2DAsync(emb_, (size_t)d * P1 * 4, e1.p, (size_t)e1.bs * 4, (size_t)d * P1 * 4, B,
                                      hipMemcpyDeviceToDevice, s));
    run_stage(0, e1, H, W);
    conv3(down12_, e1, e2, B, H, W, EP_UNSHUFFLE, TV{}, s);
    run_stage(1, e2, H2, W2);
    conv3(down23_, e2, x3, B, H2, W2, EP_UNSHUFFLE, TV{}, s);
    run_stage(2, x3, H3, W3);
    run_stage(3, x3, H3, W3);
    conv3(up32_, x3, cat2, B, H3, W3, EP_SHUFFLE, TV{}, s);
    gemm(red2_, cat2, x2, B, (int)P2, EP_STORE, TV{}, s);
    run_stage(4, x2, H2, W2);
    conv3(up21_, x2, cat1, B, H2, W2, EP_SHUFFLE, TV{}, s);
    run_stage(5, cat1, H, W);
    run_stage(6, cat1, H, W);
    if (dual_pixel_) {
      gemm(skip_, tv(emb_, (long)d * P1), cat1, B, (int)P1, EP_RESID, cat1, s);
      conv3(out_, cat1, yout, B, H, W, EP_STORE, TV{}, s);
    } else {
      conv3(out_, cat1, yout, B, H, W, EP_RESID, xin, s);
    }
  }

  int inp_ch_, out_ch_, dim_, refine_;
  std::vector<int> nb_, heads_;
  double ffn_;
  bool bias_, with_bias_ln_, dual_pixel_;
  std::vector<Ast2Stage> stages_;
  WeightPool pool_;
  PackedW patch_, down12_, down23_, up32_, up21_, red2_, skip_, out_;
  float* splitk_ = nullptr;
  DeviceBuffer arena_;
  int pB_ = 0, pH_ = 0, pW_ = 0;
  float *in_ = nullptr, *outb_ = nullptr, *cat1_ = nullptr, *cat2_ = nullptr, *x2_ = nullptr, *x3_ = nullptr;
  float *emb_ = nullptr, *t1_ = nullptr, *t2_ = nullptr, *mu_ = nullptr, *rstd_ = nullptr;
  size_t t1_per_ = 0, t2_per_ = 0;
  void* attn_scratch_ = nullptr;
};

}  // namespace

std::unique_ptr<Net> make_astv2_net(const Config& cfg) { return std::make_unique<Astv2Net>(cfg); }

}  // namespace jirf
