// AST forward graph (reference basicsr/models/archs/ast_arch.py:940-1256) assembled from the
// libjirf_hip kernels: sparse window attention (ASSA) in the bottleneck and decoder (the class hard-codes
// which layers attend and that they use the sparse form, :1017-1197), token MLP by `token_mlp`:
// "frfn" (the shipped configuration, :453-513), "leff" (the constructor default, :407-450) or
// "ffn" / "mlp" (:378-404).  Host code only.
//
// The reference carries tokens as [B, H*W, C] and converts to NCHW around every convolution; here
// every tensor stays planar [C][H*W] (the same memory as NCHW), so those transposes vanish:
// nn.LayerNorm over C is the per-pixel channel LayerNorm, nn.Linear is a 1x1 GEMM, roll / window
// partition / reverse are addressing inside the attention kernel.
//
// Buffer plan (per image, E = embed_dim, P_l = (H >> l) * (W >> l)):
//   cat_l [2 * E * 2^l][P_l], l = 0..3: upper half = encoder level l (in place), lower half =
//          the transposed-conv upsample from below; decoder level runs on the whole buffer
//   x4 [16E][P_4]  bottleneck
//   t1 (8C channels), t2 (4C channels): block temporaries
#include <algorithm>
#include <cmath>

#include "runtime.h"

namespace jirf {
namespace {

struct AstBlock {
  std::string name;
  int C = 0, hidden = 0, heads = 1, shift = 0;
  bool att = false;
  const float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;
  PackedW qkv, proj, lin1, lin2, pconv;
  const float *dw_w = nullptr, *dw_b = nullptr, *table = nullptr;
  float w_softmax = 0.5f, w_relu2 = 0.5f, scale = 1.f;
};
struct AstLayer {
  std::string name;
  int level;  // resolution level 0..4
  std::vector<AstBlock> blocks;
};

class AstNet final : public Net {
 public:
  explicit AstNet(const Config& c) {
    arch = "AST";
    cfg = c;  // AST.__init__ takes **kwargs: unknown options are ignored like in the reference
    img_size_ = cfg.get_int("img_size", 256);
    in_chans_ = cfg.get_int("in_chans", 3);
    dd_in_ = cfg.get_int("dd_in", 3);
    E_ = cfg.get_int("embed_dim", 32);
    depths_ = cfg.get_ints("depths", {2, 2, 2, 2, 2, 2, 2, 2, 2});
    heads_ = cfg.get_ints("num_heads", {1, 2, 4, 8, 16, 16, 8, 4, 2});
    win_ = cfg.get_int("win_size", 8);
    mlp_ratio_ = cfg.get_num("mlp_ratio", 4.0);
    qkv_bias_ = cfg.get_bool("qkv_bias", true);
    shift_flag_ = cfg.get_bool("shift_flag", true);
    has_qk_scale_ = cfg.has("qk_scale") && cfg.obj.at("qk_scale").kind == JsonValue::NUM;
    qk_scale_ = has_qk_scale_ ? cfg.get_num("qk_scale", 1.0) : 0.0;
    const std::string mlp = cfg.get_str("token_mlp", "leff");
    const std::string proj = cfg.get_str("token_projection", "linear");
    JIRF_REQUIRE(mlp == "frfn" || mlp == "leff" || mlp == "ffn" || mlp == "mlp", JIRF_EINVAL, "FFN error!");
    mlp_kind_ = mlp == "frfn" ? MLP_FRFN : mlp == "leff" ? MLP_LEFF : MLP_PLAIN;
    JIRF_REQUIRE(proj == "linear", JIRF_EINVAL, "Unsupported token projection!");
    JIRF_REQUIRE(depths_.size() == 9 && heads_.size() == 9, JIRF_EINVAL, "AST: depths and num_heads need 9 entries");
    JIRF_REQUIRE(E_ > 0 && E_ % 4 == 0, JIRF_EINVAL, "AST: embed_dim=%d must be a positive multiple of 4", E_);
    register_params();
  }

  std::string canonical_key(const std::string& key_in) const override {
    std::string key = Net::canonical_key(key_in);
    // a registered buffer that the kernels recompute from the window geometry
    const std::string tail = ".attn.relative_position_index";
    if (key.size() > tail.size() && key.compare(key.size() - tail.size(), tail.size(), tail) == 0) return "";
    return key;
  }

  void finalize() override {
    for (auto& p : params)
      JIRF_REQUIRE(p.is_set || p.optional, JIRF_ESTATE, "finalize: parameter '%s' was never set", p.key.c_str());
    auto vec = [&](const std::string& k) { return pool_.upload(host(k)); };
    for (auto& ly : layers_)
      for (auto& bl : ly.blocks) {
        const std::string& n = bl.name;
        const int C = bl.C, h = bl.hidden;
        if (bl.att) {
          bl.n1w = vec(n + ".norm1.weight");
          bl.n1b = vec(n + ".norm1.bias");
          const std::string a = n + ".attn";
          // to_q [C,C] and to_kv [2C,C] fused into one [3C,C] projection: rows q | k | v
          std::vector<float> w(host(a + ".qkv.to_q.weight"));
          const auto& wkv = host(a + ".qkv.to_kv.weight");
          w.insert(w.end(), wkv.begin(), wkv.end());
          std::vector<float> b;
          if (qkv_bias_) {
            b = host(a + ".qkv.to_q.bias");
            const auto& bkv = host(a + ".qkv.to_kv.bias");
            b.insert(b.end(), bkv.begin(), bkv.end());
          }
          // norm1 (weight and bias) folded into the projection: the GEMM's loader only normalises
          bl.qkv = pack_upload_ln_1x1(pool_, w.data(), qkv_bias_ ? b.data() : nullptr, host(n + ".norm1.weight").data(),
                                      host(n + ".norm1.bias").data(), 3 * C, C);
          bl.proj = pack_upload_1x1(pool_, host(a + ".proj.weight").data(), host(a + ".proj.bias").data(), C, C);
          bl.table = vec(a + ".relative_position_bias_table");
          const auto& wv = host(a + ".w");
          const float e0 = std::exp(wv[0]), e1 = std::exp(wv[1]);
          bl.w_softmax = e0 / (e0 + e1);
          bl.w_relu2 = e1 / (e0 + e1);
          bl.scale = has_qk_scale_ ? (float)qk_scale_ : 1.0f / std::sqrt((float)(C / bl.heads));
        }
        bl.n2w = vec(n + ".norm2.weight");
        bl.n2b = vec(n + ".norm2.bias");
        const std::string m = n + ".mlp";
        if (mlp_kind_ == MLP_PLAIN) {
          bl.lin1 = pack_upload_1x1(pool_, host(m + ".fc1.weight").data(), host(m + ".fc1.bias").data(), h, C);
          bl.lin2 = pack_upload_1x1(pool_, host(m + ".fc2.weight").data(), host(m + ".fc2.bias").data(), C, h);
          continue;
        }
        const int h1 = mlp_kind_ == MLP_FRFN ? 2 * h : h;   // FRFN's linear1 feeds a gate: two halves
        bl.lin1 = pack_upload_1x1(pool_, host(m + ".linear1.0.weight").data(), host(m + ".linear1.0.bias").data(), h1, C);
        bl.lin2 = pack_upload_1x1(pool_, host(m + ".linear2.0.weight").data(), host(m + ".linear2.0.bias").data(), C, h);
        if (mlp_kind_ == MLP_FRFN)
          bl.pconv = pack_upload_conv(pool_, host(m + ".partial_conv3.weight").data(), nullptr, C / 4, C / 4, 3);
        bl.dw_w = vec(m + ".dwconv.0.weight");
        bl.dw_b = vec(m + ".dwconv.0.bias");
      }
    in_proj_ = pack_upload_conv(pool_, host("input_proj.proj.0.weight").data(), host("input_proj.proj.0.bias").data(), E_, dd_in_, 3);
    out_proj_ = pack_upload_conv(pool_, host("output_proj.proj.0.weight").data(), host("output_proj.proj.0.bias").data(),
                                 in_chans_, 2 * E_, 3);
    const int dci[4] = {E_, 2 * E_, 4 * E_, 8 * E_};
    for (int i = 0; i < 4; ++i) {
      const std::string k = "dowsample_" + std::to_string(i) + ".conv.0";
      down_[i] = pack_upload_conv(pool_, host(k + ".weight").data(), host(k + ".bias").data(), 2 * dci[i], dci[i], 4);
    }
    const int uci[4] = {16 * E_, 16 * E_, 8 * E_, 4 * E_}, uco[4] = {8 * E_, 4 * E_, 2 * E_, E_};
    for (int i = 0; i < 4; ++i) {
      // ConvTranspose2d(k=2, s=2), weight [Cin][Cout][2][2]: a GEMM with rows m = 4*o + 2*i + j whose
      // output is stored like a pixel shuffle
      const std::string k = "upsample_" + std::to_string(i) + ".deconv.0";
      const auto& w = host(k + ".weight");
      const auto& b = host(k + ".bias");
      const int ci = uci[i], co = uco[i];
      std::vector<float> w2((size_t)4 * co * ci), b4((size_t)4 * co);
      for (int c = 0; c < ci; ++c)
        for (int o = 0; o < co; ++o)
          for (int t = 0; t < 4; ++t) w2[(size_t)(4 * o + t) * ci + c] = w[((size_t)c * co + o) * 4 + t];
      for (int m = 0; m < 4 * co; ++m) b4[m] = b[m >> 2];
      up_[i] = pack_upload_1x1(pool_, w2.data(), b4.data(), 4 * co, ci);
    }
    finalized = true;
  }

  size_t plan(int B, int H, int W, int mode) override {
    JIRF_REQUIRE(finalized, JIRF_ESTATE, "plan before finalize");
    JIRF_REQUIRE(B > 0 && H > 0 && W > 0, JIRF_EINVAL, "plan: bad shape B=%d H=%d W=%d", B, H, W);
    // every module of the reference derives H = W = int(sqrt(L)) (ast_arch.py:573,592,668,770) and
    // windows of 8 must tile the 1/16-resolution bottleneck
    JIRF_REQUIRE(H == W && H % 128 == 0, JIRF_EINVAL,
                 "AST needs a square input whose side is a multiple of 128, got %dx%d", H, W);
    JIRF_REQUIRE(mode == JIRF_BATCH_PER_IMAGE || mode == JIRF_BATCH_COUPLED, JIRF_EINVAL, "bad batch mode %d", mode);
    if (B == pB_ && H == pH_ && W == pW_) return arena_.bytes;
    Arena m;
    carve(m, B, H, W);
    arena_.ensure(m.off);
    Arena a;
    a.base = static_cast<unsigned char*>(arena_.p);
    carve(a, B, H, W);
    pB_ = B; pH_ = H; pW_ = W;
    stage_in = in_; stage_out = outb_; io_in_ch = dd_in_; io_out_ch = in_chans_;
    return arena_.bytes;
  }

  void forward(const float* x, float* y, int B, int H, int W, int mode, hipStream_t s) override {
    plan(B, H, W, mode);
    set_current_clock(&clock);
    clock.begin_forward();
    try {
      run(x, y, B, H, W, mode, s);
    } catch (...) {
      set_current_clock(nullptr);
      throw;
    }
    set_current_clock(nullptr);
  }

 private:
  void register_params() {
    static const char* names[9] = {"encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
                                   "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3"};
    const int mult[9] = {1, 2, 4, 8, 16, 16, 8, 4, 2};
    const int level[9] = {0, 1, 2, 3, 4, 3, 2, 1, 0};
    add_param("input_proj.proj.0.weight", {E_, dd_in_, 3, 3});
    add_param("input_proj.proj.0.bias", {E_});
    add_param("output_proj.proj.0.weight", {in_chans_, 2 * E_, 3, 3});
    add_param("output_proj.proj.0.bias", {in_chans_});
    for (int li = 0; li < 9; ++li) {
      AstLayer ly;
      ly.name = names[li];
      ly.level = level[li];
      for (int i = 0; i < depths_[li]; ++i) {
        AstBlock bl;
        bl.name = ly.name + ".blocks." + std::to_string(i);
        bl.C = E_ * mult[li];
        bl.hidden = (int)((double)bl.C * mlp_ratio_);
        bl.heads = heads_[li];
        bl.att = li >= 4;
        // TransformerBlock.__init__ (ast_arch.py:712-714): the window shrinks to the layer's NOMINAL
        // resolution (img_size >> level) when that is <= win_size, and such a block never shifts
        int win = win_;
        int shift = (shift_flag_ && (i % 2 == 1)) ? win_ / 2 : 0;
        const int nominal = img_size_ >> level[li];
        if (nominal <= win_) { win = nominal; shift = 0; }
        bl.shift = shift;
        if (bl.att) {
          JIRF_REQUIRE(win == 8, JIRF_EINVAL, "AST: window %d at %s (only 8x8 windows are built)", win, bl.name.c_str());
          JIRF_REQUIRE(bl.C % bl.heads == 0 && bl.C / bl.heads == 32, JIRF_EINVAL,
                       "AST: head dim %d at %s (only head_dim 32 is built)", bl.C / std::max(bl.heads, 1), bl.name.c_str());
        }
        register_block(bl);
        ly.blocks.push_back(std::move(bl));
      }
      layers_.push_back(std::move(ly));
    }
    const int dci[4] = {E_, 2 * E_, 4 * E_, 8 * E_};
    for (int i = 0; i < 4; ++i) {
      const std::string k = "dowsample_" + std::to_string(i) + ".conv.0";
      add_param(k + ".weight", {2 * dci[i], dci[i], 4, 4});
      add_param(k + ".bias", {2 * dci[i]});
    }
    const int uci[4] = {16 * E_, 16 * E_, 8 * E_, 4 * E_}, uco[4] = {8 * E_, 4 * E_, 2 * E_, E_};
    for (int i = 0; i < 4; ++i) {
      const std::string k = "upsample_" + std::to_string(i) + ".deconv.0";
      add_param(k + ".weight", {uci[i], uco[i], 2, 2});
      add_param(k + ".bias", {uco[i]});
    }
  }

  void register_block(const AstBlock& bl) {
    const std::string& n = bl.name;
    const int C = bl.C, h = bl.hidden;
    if (bl.att) {
      add_param(n + ".norm1.weight", {C});
      add_param(n + ".norm1.bias", {C});
      const std::string a = n + ".attn";
      add_param(a + ".relative_position_bias_table", {225, bl.heads});
      add_param(a + ".qkv.to_q.weight", {C, C});
      if (qkv_bias_) add_param(a + ".qkv.to_q.bias", {C});
      add_param(a + ".qkv.to_kv.weight", {2 * C, C});
      if (qkv_bias_) add_param(a + ".qkv.to_kv.bias", {2 * C});
      add_param(a + ".proj.weight", {C, C});
      add_param(a + ".proj.bias", {C});
      add_param(a + ".w", {2});
    }
    add_param(n + ".norm2.weight", {C});
    add_param(n + ".norm2.bias", {C});
    const std::string m = n + ".mlp";
    if (mlp_kind_ == MLP_PLAIN) {             // Mlp (ast_arch.py:378-404)
      add_param(m + ".fc1.weight", {h, C});
      add_param(m + ".fc1.bias", {h});
      add_param(m + ".fc2.weight", {C, h});
      add_param(m + ".fc2.bias", {C});
      return;
    }
    const int h1 = mlp_kind_ == MLP_FRFN ? 2 * h : h;
    add_param(m + ".linear1.0.weight", {h1, C});
    add_param(m + ".linear1.0.bias", {h1});
    add_param(m + ".dwconv.0.weight", {h, 1, 3, 3});
    add_param(m + ".dwconv.0.bias", {h});
    add_param(m + ".linear2.0.weight", {C, h});
    add_param(m + ".linear2.0.bias", {C});
    if (mlp_kind_ == MLP_FRFN) add_param(m + ".partial_conv3.weight", {C / 4, C / 4, 3, 3});
  }

  void carve(Arena& a, int B, int H, int W) {
    splitk_ = a.floats(SPLITK_WS_FLOATS);   // slabs of split-K GEMMs (launch_gemm)
    const size_t nB = B, E = E_;
    size_t P[5];
    for (int l = 0; l < 5; ++l) P[l] = (size_t)(H >> l) * (W >> l);
    in_ = a.floats(nB * dd_in_ * P[0]);
    outb_ = a.floats(nB * in_chans_ * P[0]);
    for (int l = 0; l < 4; ++l) cat_[l] = a.floats(nB * 2 * (E << l) * P[l]);
    x4_ = a.floats(nB * 16 * E * P[4]);
    size_t t1 = 0, t2 = 0;
    for (auto& ly : layers_)
      for (auto& bl : ly.blocks) {
        t1 = std::max(t1, (size_t)std::max(3 * bl.C, 2 * bl.hidden) * P[ly.level]);
        t2 = std::max(t2, (size_t)std::max(bl.hidden, bl.C + bl.C / 4) * P[ly.level]);
      }
    t1_ = a.floats(nB * t1);
    t2_ = a.floats(nB * t2);
    t1_per_ = t1; t2_per_ = t2;
    mu_ = a.floats(nB * P[0]);
    rstd_ = a.floats(nB * P[0]);
  }

  void gemm(const PackedW& w, TV x, TV y, int B, int P, int epi, TV r, hipStream_t s) {
    GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
    g.B = B; g.M = w.M; g.K = w.K; g.P = P; g.Mpad = w.Mpad;
    g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.bias = w.bias; g.x = x; g.y = y; g.epi = epi; g.r = r; g.imgs_per_group = 1;
    launch_gemm(g, LD_PLAIN, s);
  }
  void conv(const PackedW& w, TV x, TV y, int B, int Hin, int Win, int ks, int stride, int pad, int epi, TV r,
            hipStream_t s) {
    GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
    g.H = (Hin + 2 * pad - ks) / stride + 1;
    g.W = (Win + 2 * pad - ks) / stride + 1;
    g.B = B; g.M = w.M; g.K = w.K; g.P = g.H * g.W; g.Mpad = w.Mpad;
    g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.bias = w.bias; g.x = x; g.y = y; g.epi = epi; g.r = r;
    g.Cin = w.Cin; g.ks = ks; g.stride = stride; g.pad = pad; g.Hin = Hin; g.Win = Win; g.imgs_per_group = 1;
    launch_gemm(g, LD_CONV3, s);
  }

  void run_block(const AstBlock& bl, TV x, int B, int H, int W, int mode, hipStream_t s) {
    const int C = bl.C, P = H * W, h = bl.hidden;
    const TV t1 = tv(t1_, (long)t1_per_), t2 = tv(t2_, (long)t2_per_);
    if (bl.att) {
      // norm1 -> fused q|k|v projection -> window attention -> proj + shortcut
      if (!gemm_ln_inline(C, 3 * C, P, B)) launch_ln_stats(x, mu_, rstd_, B, C, P, s);
      GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
      g.B = B; g.M = 3 * C; g.K = C; g.P = P; g.Mpad = bl.qkv.Mpad; g.wt = bl.qkv.wt; g.bias = bl.qkv.bias;
      g.wt3 = bl.qkv.wt3; g.wt3_plane = bl.qkv.wt3_plane;
      g.x = x; g.y = t1; g.epi = EP_STORE; g.imgs_per_group = 1;
      g.mu = mu_; g.rstd = rstd_; g.lnw = nullptr; g.lnb = nullptr; g.ln_center = 1;
      launch_gemm(g, LD_LN, s);
      WinAttnArgs wa{};
      wa.B = B; wa.C = C; wa.H = H; wa.W = W; wa.heads = bl.heads; wa.shift = bl.shift;
      wa.coupled = mode == JIRF_BATCH_COUPLED ? 1 : 0;
      wa.q = tv_chan(t1, 0, P); wa.k = tv_chan(t1, C, P); wa.v = tv_chan(t1, 2 * C, P);
      wa.out = tv_chan(t2, 0, P);
      wa.table = bl.table; wa.scale = bl.scale; wa.w_softmax = bl.w_softmax; wa.w_relu2 = bl.w_relu2;
      launch_window_attention(wa, s);
      gemm(bl.proj, tv_chan(t2, 0, P), x, B, P, EP_RESID, x, s);
    }
    launch_layernorm(x, bl.n2w, bl.n2b, tv_chan(t2, 0, P), B, C, P, 1, s);
    if (mlp_kind_ == MLP_PLAIN) {
      // Mlp (ast_arch.py:396-402): fc1 + GELU -> fc2 (dropout is the identity in eval)
      gemm(bl.lin1, tv_chan(t2, 0, P), t1, B, P, EP_GELU, TV{}, s);
      gemm(bl.lin2, t1, x, B, P, EP_RESID, x, s);
      return;
    }
    if (mlp_kind_ == MLP_LEFF) {
      // LeFF (ast_arch.py:425-450): linear1 + GELU -> depthwise 3x3 + GELU -> linear2; eca is the identity
      gemm(bl.lin1, tv_chan(t2, 0, P), t1, B, P, EP_GELU, TV{}, s);
      launch_dwconv3x3(t1, bl.dw_w, bl.dw_b, t2, B, h, H, W, 4, s);
      gemm(bl.lin2, t2, x, B, P, EP_RESID, x, s);
      return;
    }
    // FRFN: norm2 -> dense 3x3 on the first C/4 channels -> linear1 + GELU -> gelu(dw3x3(x1)) * x2 -> linear2
    conv(bl.pconv, tv_chan(t2, 0, P), tv_chan(t2, C, P), B, H, W, 3, 1, 1, EP_STORE, TV{}, s);
    {
      GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
      g.B = B; g.M = 2 * h; g.K = C; g.P = P; g.Mpad = bl.lin1.Mpad; g.wt = bl.lin1.wt; g.bias = bl.lin1.bias;
      g.wt3 = bl.lin1.wt3; g.wt3_plane = bl.lin1.wt3_plane;
      g.x = tv_chan(t2, 0, P); g.x2 = tv_chan(t2, C, P); g.k_split = C / 4;
      g.y = t1; g.epi = EP_GELU; g.imgs_per_group = 1;
      launch_gemm(g, LD_PLAIN, s);
    }
    launch_dwconv3x3(t1, bl.dw_w, bl.dw_b, t2, B, 2 * h, H, W, 3, s);
    gemm(bl.lin2, t2, x, B, P, EP_RESID, x, s);
  }

  void run(const float* x, float* y, int B, int H, int W, int mode, hipStream_t s) {
    const int E = E_;
    long P[5];
    int Hl[5], Wl[5];
    for (int l = 0; l < 5; ++l) { Hl[l] = H >> l; Wl[l] = W >> l; P[l] = (long)Hl[l] * Wl[l]; }
    const TV xin = tv(const_cast<float*>(x), (long)dd_in_ * P[0]);
    const TV yout = tv(y, (long)in_chans_ * P[0]);
    TV cat[4], enc[4];
    for (int l = 0; l < 4; ++l) {
      cat[l] = tv(cat_[l], 2L * (E << l) * P[l]);
      enc[l] = tv_chan(cat[l], E << l, P[l]);
    }
    const TV x4 = tv(x4_, 16L * E * P[4]);
    auto run_layer = [&](int idx, TV t) {
      const int l = layers_[idx].level;
      for (auto& bl : layers_[idx].blocks) run_block(bl, t, B, Hl[l], Wl[l], mode, s);
    };
    conv(in_proj_, xin, enc[0], B, H, W, 3, 1, 1, EP_LEAKY, TV{}, s);
    for (int l = 0; l < 4; ++l) {
      run_layer(l, enc[l]);
      conv(down_[l], enc[l], l < 3 ? enc[l + 1] : x4, B, Hl[l], Wl[l], 4, 2, 1, EP_STORE, TV{}, s);
    }
    run_layer(4, x4);
    TV below = x4;
    for (int i = 0; i < 4; ++i) {
      const int l = 3 - i;  // upsample_i: level l+1 -> level l, into the lower half of cat_l
      GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
      g.B = B; g.M = up_[i].M; g.K = up_[i].K; g.P = (int)P[l + 1]; g.Mpad = up_[i].Mpad;
      g.wt = up_[i].wt; g.bias = up_[i].bias; g.x = below; g.y = cat[l];
      g.epi = EP_SHUFFLE; g.H = Hl[l + 1]; g.W = Wl[l + 1]; g.imgs_per_group = 1;
      launch_gemm(g, LD_PLAIN, s);
      run_layer(5 + i, cat[l]);
      below = cat[l];
    }
    conv(out_proj_, cat[0], yout, B, H, W, 3, 1, 1, dd_in_ == 3 ? EP_RESID : EP_STORE, dd_in_ == 3 ? xin : TV{}, s);
  }

  enum { MLP_FRFN = 0, MLP_LEFF = 1, MLP_PLAIN = 2 };
  int mlp_kind_ = MLP_FRFN;
  int img_size_, in_chans_, dd_in_, E_, win_;
  std::vector<int> depths_, heads_;
  double mlp_ratio_, qk_scale_;
  bool qkv_bias_, shift_flag_, has_qk_scale_;
  std::vector<AstLayer> layers_;
  WeightPool pool_;
  PackedW in_proj_, out_proj_, down_[4], up_[4];
  float* splitk_ = nullptr;
  DeviceBuffer arena_;
  int pB_ = 0, pH_ = 0, pW_ = 0;
  float *in_ = nullptr, *outb_ = nullptr, *cat_[4] = {nullptr, nullptr, nullptr, nullptr}, *x4_ = nullptr;
  float *t1_ = nullptr, *t2_ = nullptr, *mu_ = nullptr, *rstd_ = nullptr;
  size_t t1_per_ = 0, t2_per_ = 0;
};

}  // namespace

std::unique_ptr<Net> make_ast_net(const Config& cfg) { return std::make_unique<AstNet>(cfg); }

}  // namespace jirf
