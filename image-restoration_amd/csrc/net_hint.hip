// HINT forward graph (reference basicsr/models/archs/hint_arch.py:494-706) assembled from the
// libjirf_hip kernels.  Host code only: parameter registry, weight packing, arena planning and
// the launch sequence.  Nothing here allocates inside forward().
//
// Buffer plan (per image; P1 = H*W, P2 = P1/4, P3 = P1/16, P4 = P1/64, d = dim):
//   cat1 [2d][P1]  channels [d,2d) hold encoder level 1 (patch embed + its blocks run there in
//                  place); up2_1 later fills [0,d) and decoder_level1 / refinement run on all 2d.
//   cat2 [4d][P2], cat3 [8d][P3]  same trick for levels 2 and 3 (so torch.cat never happens)
//   x2 [2d][P2], x3 [4d][P3]      outputs of reduce_chan_level{2,3}; decoder levels run in place
//   x4 [8d][P4]                   latent
//   t1, t2                        block temporaries (qkv / dw outputs / GDFN hidden)
#include <algorithm>
#include <cmath>
#include <cstring>

#include "runtime.h"

namespace jirf {
namespace {

struct LNW {
  const float* w = nullptr;
  const float* b = nullptr;
};
struct DWW {
  const float* w = nullptr;
  const float* b = nullptr;
};

struct HintBlock {
  std::string name;
  int C = 0, hidden = 0;
  bool att = false;
  LNW n1, n2;
  PackedW qkv, gate, tail, pin, pout;   // tail: project_out folded with the intra-modulator's down/up
  DWW qkv_dw, ffn_dw;
  HintAttnWeights aw{};
};

struct Stage {
  std::string name;
  int level;  // 1..4 (resolution level)
  std::vector<HintBlock> blocks;
};

class HintNet final : public Net {
 public:
  explicit HintNet(const Config& c) {
    arch = "HINT";
    cfg = c;
    static const char* known[] = {"inp_channels", "out_channels", "dim", "num_blocks",
                                  "num_refinement_blocks", "heads", "ffn_expansion_factor", "bias",
                                  "ln_type", "dual_pixel_task", "qv_cache"};
    for (auto& kv : cfg.obj) {
      bool ok = false;
      for (auto k : known) ok = ok || kv.first == k;
      // HINT.__init__ has no **kwargs: an unknown option is a TypeError in the reference
      JIRF_REQUIRE(ok, JIRF_EINVAL, "HINT got an unexpected keyword argument '%s'", kv.first.c_str());
    }
    inp_ch_ = cfg.get_int("inp_channels", 3);
    out_ch_ = cfg.get_int("out_channels", 3);
    dim_ = cfg.get_int("dim", 48);
    nb_ = cfg.get_ints("num_blocks", {4, 6, 6, 8});
    refine_ = cfg.get_int("num_refinement_blocks", 4);
    ffn_ = cfg.get_num("ffn_expansion_factor", 2.66);
    bias_ = cfg.get_bool("bias", false);
    with_bias_ln_ = cfg.get_str("ln_type", "WithBias") != "BiasFree";
    dual_pixel_ = cfg.get_bool("dual_pixel_task", false);
    JIRF_REQUIRE(nb_.size() >= 3, JIRF_EINVAL, "HINT: num_blocks needs at least 3 entries");
    JIRF_REQUIRE(dim_ > 0 && dim_ % 8 == 0, JIRF_EINVAL,
                 "HINT: dim=%d must be a positive multiple of 8 (channel groups of C/8)", dim_);
    JIRF_REQUIRE(inp_ch_ > 0 && out_ch_ > 0, JIRF_EINVAL, "HINT: bad channel counts");
    JIRF_REQUIRE(dual_pixel_ || inp_ch_ == out_ch_, JIRF_EINVAL,
                 "HINT: the output residual needs inp_channels == out_channels");
    register_params();
  }

  std::string canonical_key(const std::string& key_in) const override {
    std::string key = Net::canonical_key(key_in);
    if (key == "qv_cache") return "";  // every test script deletes it before loading
    // the four inter-modulators are also reachable through the Sequential (hint_arch.py:325-332)
    const std::string tag = ".attn.inter_modulators.";
    auto pos = key.find(tag);
    if (pos != std::string::npos) {
      const size_t d = pos + tag.size();
      if (d < key.size() && key[d] >= '0' && key[d] <= '3')
        key = key.substr(0, pos) + ".attn.inter_modulator" + char(key[d] + 1) + key.substr(d + 1);
    }
    return key;
  }

  void finalize() override {
    for (auto& p : params)
      JIRF_REQUIRE(p.is_set || p.optional, JIRF_ESTATE, "finalize: parameter '%s' was never set",
                   p.key.c_str());
    auto conv1 = [&](const std::string& k, int M, int K) {
      return pack_upload_1x1(pool_, host(k + ".weight").data(), bias_ ? host(k + ".bias").data() : nullptr, M, K);
    };
    auto conv1_ln = [&](const std::string& k, const std::string& ln, int M, int K) {  // LayerNorm folded in
      return pack_upload_ln_1x1(pool_, host(k + ".weight").data(), bias_ ? host(k + ".bias").data() : nullptr,
                                host(ln + ".body.weight").data(),
                                with_bias_ln_ ? host(ln + ".body.bias").data() : nullptr, M, K);
    };
    auto conv1b = [&](const std::string& k, int M, int K) {  // Conv1d(k=1), always biased
      return pack_upload_1x1(pool_, host(k + ".weight").data(), host(k + ".bias").data(), M, K);
    };
    auto conv3 = [&](const std::string& k, int M, int Cin, bool b) {
      return pack_upload_3x3(pool_, host(k + ".weight").data(), b ? host(k + ".bias").data() : nullptr, M, Cin);
    };
    auto vec = [&](const std::string& k) { return pool_.upload(host(k)); };
    // Everything after the gate of the intra-modulator is linear (hint_arch.py:226-228,401,
    // :403 project_out):  y = x + Wp (Wu (Wd g + bd) + bu + out) + bp  with g the gated tensor and
    // out = attention * V.  Folded in double precision into ONE 1x1 conv over the stacked input
    // [g; out]:  W = [Wp Wu Wd | Wp],  b = Wp (Wu bd + bu) + bp  -- three HBM-bound pointwise
    // GEMMs and two intermediate tensors fewer per block.
    auto fold_attention_tail = [&](const std::string& a, int C) {
      const int Ch = C / 2;
      const std::vector<float>& Wp = host(a + ".project_out.weight");              // [C][C]
      const std::vector<float>& Wd = host(a + ".intra_modulator.down.weight");      // [Ch][C]
      const std::vector<float>& bd = host(a + ".intra_modulator.down.bias");
      const std::vector<float>& Wu = host(a + ".intra_modulator.up.weight");        // [C][Ch]
      const std::vector<float>& bu = host(a + ".intra_modulator.up.bias");
      std::vector<double> ud((size_t)C * C), bud(C);
      for (int i = 0; i < C; ++i) {
        for (int j = 0; j < C; ++j) {
          double t = 0.0;
          for (int m = 0; m < Ch; ++m) t += (double)Wu[(size_t)i * Ch + m] * (double)Wd[(size_t)m * C + j];
          ud[(size_t)i * C + j] = t;
        }
        double t = bu[i];
        for (int m = 0; m < Ch; ++m) t += (double)Wu[(size_t)i * Ch + m] * (double)bd[m];
        bud[i] = t;
      }
      std::vector<float> W((size_t)C * 2 * C), b(C);
      for (int i = 0; i < C; ++i) {
        for (int j = 0; j < C; ++j) {
          double t = 0.0;
          for (int m = 0; m < C; ++m) t += (double)Wp[(size_t)i * C + m] * ud[(size_t)m * C + j];
          W[(size_t)i * 2 * C + j] = (float)t;
          W[(size_t)i * 2 * C + C + j] = Wp[(size_t)i * C + j];
        }
        double t = bias_ ? (double)host(a + ".project_out.bias")[i] : 0.0;
        for (int m = 0; m < C; ++m) t += (double)Wp[(size_t)i * C + m] * bud[m];
        b[i] = (float)t;
      }
      return pack_upload_1x1(pool_, W.data(), b.data(), C, 2 * C);
    };
    for (auto& st : stages_) {
      for (auto& bl : st.blocks) {
        const std::string& n = bl.name;
        const int C = bl.C, h = bl.hidden;
        if (bl.att) {
          bl.n1.w = vec(n + ".norm1.body.weight");
          if (with_bias_ln_) bl.n1.b = vec(n + ".norm1.body.bias");
          const std::string a = n + ".attn";
          bl.qkv = conv1_ln(a + ".qkv", n + ".norm1", 3 * C, C);
          bl.qkv_dw.w = vec(a + ".qkv_dwconv.weight");
          if (bias_) bl.qkv_dw.b = vec(a + ".qkv_dwconv.bias");
          bl.gate = conv1b(a + ".intra_modulator.gating_conv", C, C);
          bl.tail = fold_attention_tail(a, C);
          bl.aw.temperature = vec(a + ".temperature");
          for (int g = 0; g < 4; ++g) {
            const std::string m = a + ".inter_modulator" + std::to_string(g + 1);
            const int cg = (int)host(m + ".gating_conv.bias").size();
            std::vector<float> packed;
            pack_hint_modulator(host(m + ".conv_width.weight").data(), 2 * cg, cg, packed);
            bl.aw.cw_wt[g] = pool_.upload(packed);
            bl.aw.cw_b[g] = vec(m + ".conv_width.bias");
            pack_hint_modulator(host(m + ".gating_conv.weight").data(), cg, cg, packed);
            bl.aw.gc_wt[g] = pool_.upload(packed);
            bl.aw.gc_b[g] = vec(m + ".gating_conv.bias");
          }
        }
        bl.n2.w = vec(n + ".norm2.body.weight");
        if (with_bias_ln_) bl.n2.b = vec(n + ".norm2.body.bias");
        bl.pin = conv1_ln(n + ".ffn.project_in", n + ".norm2", 2 * h, C);
        bl.ffn_dw.w = vec(n + ".ffn.dwconv.weight");
        if (bias_) bl.ffn_dw.b = vec(n + ".ffn.dwconv.bias");
        bl.pout = conv1(n + ".ffn.project_out", C, h);
      }
    }
    const int d = dim_;
    patch_ = conv3("patch_embed.proj", d, inp_ch_, false);
    down12_ = conv3("down1_2.body.0", d / 2, d, false);
    down23_ = conv3("down2_3.body.0", d, 2 * d, false);
    down34_ = conv3("down3_4.body.0", 2 * d, 4 * d, false);
    up43_ = conv3("up4_3.body.0", 16 * d, 8 * d, false);
    up32_ = conv3("up3_2.body.0", 8 * d, 4 * d, false);
    up21_ = conv3("up2_1.body.0", 4 * d, 2 * d, false);
    red3_ = conv1("reduce_chan_level3", 4 * d, 8 * d);
    red2_ = conv1("reduce_chan_level2", 2 * d, 4 * d);
    if (dual_pixel_) skip_ = conv1("skip_conv", 2 * d, d);
    out_ = conv3("output", out_ch_, 2 * d, bias_);
    finalized = true;
  }

  size_t plan(int B, int H, int W, int mode) override {
    JIRF_REQUIRE(finalized, JIRF_ESTATE, "plan before finalize");
    JIRF_REQUIRE(B > 0 && H > 0 && W > 0, JIRF_EINVAL, "plan: bad shape B=%d H=%d W=%d", B, H, W);
    // three stride-2 levels (hint_arch.py:454-471): the reference's reshape fails otherwise
    JIRF_REQUIRE(H % 8 == 0 && W % 8 == 0, JIRF_EINVAL,
                 "HINT needs H and W to be multiples of 8, got %dx%d (pad the input as the test scripts do)",
                 H, W);
    JIRF_REQUIRE(mode == JIRF_BATCH_PER_IMAGE || mode == JIRF_BATCH_COUPLED, JIRF_EINVAL, "bad batch mode %d", mode);
    if (B == pB_ && H == pH_ && W == pW_ && mode == pmode_) return arena_.bytes;
    Arena m;  // measure
    carve(m, B, H, W, mode);
    arena_.ensure(m.off);
    Arena a;
    a.base = static_cast<unsigned char*>(arena_.p);
    carve(a, B, H, W, mode);
    pB_ = B; pH_ = H; pW_ = W; pmode_ = mode;
    stage_in = in_; stage_out = outb_; io_in_ch = inp_ch_; io_out_ch = out_ch_;
    ensure_cache_store(mode == JIRF_BATCH_COUPLED ? 1 : B);
    // a cache carried from a call with another batch layout cannot be continued: it is kept, and the
    // next forward reports JIRF_ESTATE until the caller resets or replaces it (never dropped silently)
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

  void reset_cache() override { cache_valid_ = false; }

  int get_cache(float* out, int* n, int* c, size_t cap) override {
    JIRF_REQUIRE(cache_valid_, JIRF_ESTATE, "qv_cache is None");
    const size_t cnt = (size_t)cache_g_ * cache_c_ * cache_c_;
    JIRF_REQUIRE(cap >= cnt, JIRF_EINVAL, "get_cache: buffer of %zu floats, need %zu", cap, cnt);
    JIRF_HIP_CHECK(hipDeviceSynchronize());
    JIRF_HIP_CHECK(hipMemcpy(out, cache_store_.p, cnt * 4, hipMemcpyDeviceToHost));
    *n = cache_g_;
    *c = cache_c_;
    return JIRF_OK;
  }
  int set_cache(const float* in, int n, int c) override {
    JIRF_REQUIRE(in && n > 0 && c > 0, JIRF_EINVAL, "set_cache: bad arguments");
    const size_t cnt = (size_t)n * c * c;
    JIRF_HIP_CHECK(hipDeviceSynchronize());
    cache_store_.ensure(std::max(cnt * 4, cache_store_.bytes));
    JIRF_HIP_CHECK(hipMemcpy(cache_store_.p, in, cnt * 4, hipMemcpyHostToDevice));
    cache_g_ = n; cache_c_ = c; cache_valid_ = true;
    return JIRF_OK;
  }

  int attention_blocks(int* n, int* chans, int cap) override {
    int cnt = 0;
    for (auto& st : stages_)
      for (auto& bl : st.blocks)
        if (bl.att) {
          if (chans && cnt < cap) chans[cnt] = bl.C;
          ++cnt;
        }
    *n = cnt;
    return JIRF_OK;
  }
  int get_perms(int32_t* out, size_t cap, size_t* n) override {
    JIRF_REQUIRE(pB_ > 0 && perm_total_ > 0, JIRF_ESTATE, "get_perms before a forward");
    *n = perm_total_;
    JIRF_REQUIRE(cap >= perm_total_, JIRF_EINVAL, "get_perms: need %zu ints", perm_total_);
    JIRF_HIP_CHECK(hipDeviceSynchronize());
    JIRF_HIP_CHECK(hipMemcpy(out, perms_, perm_total_ * 4, hipMemcpyDeviceToHost));
    return JIRF_OK;
  }
  int set_forced_perms(const int32_t* in, size_t n) override {
    if (!in) {
      forced_on_ = false;
      return JIRF_OK;
    }
    JIRF_REQUIRE(pB_ > 0, JIRF_ESTATE, "set_forced_perms needs a planned shape (call plan first)");
    JIRF_REQUIRE(n == perm_total_, JIRF_EINVAL, "set_forced_perms: got %zu ints, layout has %zu", n, perm_total_);
    // every row must be a permutation: a bad index would gather out of bounds on the device
    size_t off = 0;
    const int G = pmode_ == JIRF_BATCH_COUPLED ? 1 : pB_;
    for (auto& st : stages_)
      for (auto& bl : st.blocks) {
        if (!bl.att) continue;
        for (int g = 0; g < G; ++g) {
          std::vector<char> seen(bl.C, 0);
          for (int i = 0; i < bl.C; ++i) {
            const int v = in[off + (size_t)g * bl.C + i];
            JIRF_REQUIRE(v >= 0 && v < bl.C && !seen[v], JIRF_EINVAL,
                         "set_forced_perms: block %s row %d is not a permutation", bl.name.c_str(), g);
            seen[v] = 1;
          }
        }
        off += (size_t)G * bl.C;
      }
    JIRF_HIP_CHECK(hipDeviceSynchronize());
    JIRF_HIP_CHECK(hipMemcpy(forced_, in, n * 4, hipMemcpyHostToDevice));
    forced_on_ = true;
    return JIRF_OK;
  }

 private:
  // ---- construction ------------------------------------------------------------------------
  void register_params() {
    const int d = dim_;
    add_param("patch_embed.proj.weight", {d, inp_ch_, 3, 3});
    struct SP { const char* name; int n; int mult; bool att; int level; };
    const SP plan[] = {
        {"encoder_level1", nb_[0], 1, false, 1}, {"encoder_level2", nb_[1], 2, false, 2},
        {"encoder_level3", nb_[2], 4, false, 3}, {"latent", nb_[1], 8, true, 4},
        {"decoder_level3", nb_[2], 4, true, 3},  {"decoder_level2", nb_[1], 2, true, 2},
        {"decoder_level1", nb_[0], 2, true, 1},  {"refinement", refine_, 2, true, 1}};
    for (auto& sp : plan) {
      Stage st;
      st.name = sp.name;
      st.level = sp.level;
      for (int i = 0; i < sp.n; ++i) {
        HintBlock bl;
        bl.name = std::string(sp.name) + "." + std::to_string(i);
        bl.C = d * sp.mult;
        bl.hidden = (int)((double)bl.C * ffn_);  // int(dim * ffn_expansion_factor), hint_arch.py:154
        bl.att = sp.att;
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
    conv("down3_4.body.0", 2 * d, 4 * d, 3, false);
    conv("up4_3.body.0", 16 * d, 8 * d, 3, false);
    conv("up3_2.body.0", 8 * d, 4 * d, 3, false);
    conv("up2_1.body.0", 4 * d, 2 * d, 3, false);
    conv("reduce_chan_level3", 4 * d, 8 * d, 1, bias_);
    conv("reduce_chan_level2", 2 * d, 4 * d, 1, bias_);
    if (dual_pixel_) conv("skip_conv", 2 * d, d, 1, bias_);
    conv("output", out_ch_, 2 * d, 3, bias_);
  }

  void register_block(const HintBlock& bl) {
    const std::string& n = bl.name;
    const int C = bl.C, h = bl.hidden;
    auto ln = [&](const std::string& k) {
      add_param(k + ".body.weight", {C});
      if (with_bias_ln_) add_param(k + ".body.bias", {C});
    };
    if (bl.att) {
      ln(n + ".norm1");
      const std::string a = n + ".attn";
      add_param(a + ".temperature", {4, 1, 1});
      add_param(a + ".qkv.weight", {3 * C, C, 1, 1});
      if (bias_) add_param(a + ".qkv.bias", {3 * C});
      add_param(a + ".qkv_dwconv.weight", {3 * C, 1, 3, 3});
      if (bias_) add_param(a + ".qkv_dwconv.bias", {3 * C});
      add_param(a + ".project_out.weight", {C, C, 1, 1});
      if (bias_) add_param(a + ".project_out.bias", {C});
      const std::string im = a + ".intra_modulator";
      add_param(im + ".down.weight", {C / 2, C, 1});
      add_param(im + ".down.bias", {C / 2});
      add_param(im + ".up.weight", {C, C / 2, 1});
      add_param(im + ".up.bias", {C});
      add_param(im + ".gating_conv.weight", {C, C, 1});
      add_param(im + ".gating_conv.bias", {C});
      const int r[4] = {1, 2, 2, 3};
      for (int g = 0; g < 4; ++g) {
        const int cg = r[g] * C / 8;  // in_c = ratio * dim // 8, hint_arch.py:321-324
        const std::string m = a + ".inter_modulator" + std::to_string(g + 1);
        add_param(m + ".conv_width.weight", {2 * cg, cg, 1});
        add_param(m + ".conv_width.bias", {2 * cg});
        add_param(m + ".gating_conv.weight", {cg, cg, 1});
        add_param(m + ".gating_conv.bias", {cg});
      }
    }
    ln(n + ".norm2");
    add_param(n + ".ffn.project_in.weight", {2 * h, C, 1, 1});
    if (bias_) add_param(n + ".ffn.project_in.bias", {2 * h});
    add_param(n + ".ffn.dwconv.weight", {2 * h, 1, 3, 3});
    if (bias_) add_param(n + ".ffn.dwconv.bias", {2 * h});
    add_param(n + ".ffn.project_out.weight", {C, h, 1, 1});
    if (bias_) add_param(n + ".ffn.project_out.bias", {C});
  }

  // ---- arena -----------------------------------------------------------------------------------
  void carve(Arena& a, int B, int H, int W, int mode) {
    splitk_ = a.floats(SPLITK_WS_FLOATS);   // slabs of split-K GEMMs (launch_gemm)
    const size_t d = dim_, P1 = (size_t)H * W, P2 = P1 / 4, P3 = P1 / 16, P4 = P1 / 64;
    const size_t nB = B;
    in_ = a.floats(nB * inp_ch_ * P1);
    outb_ = a.floats(nB * out_ch_ * P1);
    cat1_ = a.floats(nB * 2 * d * P1);
    cat2_ = a.floats(nB * 4 * d * P2);
    cat3_ = a.floats(nB * 8 * d * P3);
    x2_ = a.floats(nB * 2 * d * P2);
    x3_ = a.floats(nB * 4 * d * P3);
    x4_ = a.floats(nB * 8 * d * P4);
    emb_ = dual_pixel_ ? a.floats(nB * d * P1) : nullptr;
    size_t t1 = 0, t2 = 0, scratch = 0;
    perm_total_ = 0;
    const int G = mode == JIRF_BATCH_COUPLED ? 1 : B;
    int cmax = 0;
    for (auto& st : stages_) {
      const size_t P = st.level == 1 ? P1 : st.level == 2 ? P2 : st.level == 3 ? P3 : P4;
      for (auto& bl : st.blocks) {
        t1 = std::max(t1, (size_t)std::max(3 * bl.C, 2 * bl.hidden) * P);
        t2 = std::max(t2, (size_t)std::max(3 * bl.C, bl.hidden) * P);
        if (bl.att) {
          HintAttnScratch tmp;
          scratch = std::max(scratch, hint_attn_scratch_carve(hint_attn_shape(B, bl.C, (int)P, mode), nullptr, &tmp));
          perm_total_ += (size_t)G * bl.C;
          cmax = std::max(cmax, bl.C);
        }
      }
    }
    t1_ = a.floats(nB * t1);
    t2_ = a.floats(nB * t2);
    t1_per_ = t1; t2_per_ = t2;
    mu_ = a.floats(nB * P1);
    rstd_ = a.floats(nB * P1);
    attn_scratch_ = a.take(scratch);
    perms_ = static_cast<int*>(a.take(std::max<size_t>(perm_total_, 1) * 4));
    forced_ = static_cast<int*>(a.take(std::max<size_t>(perm_total_, 1) * 4));
    cache_a_ = a.floats((size_t)G * cmax * cmax);
    cache_b_ = a.floats((size_t)G * cmax * cmax);
    cmax_ = cmax;
    forced_on_ = false;  // a new layout invalidates pinned orders
  }

  // ---- launch helpers ---------------------------------------------------------------------------
  void gemm(const PackedW& w, TV x, TV y, int B, int P, int epi, TV r, hipStream_t s,
            int loader = LD_PLAIN, const LNW* ln = nullptr) {
    GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
    g.B = B; g.M = w.M; g.K = w.K; g.P = P; g.Mpad = w.Mpad;
    g.wt = w.wt; g.wt3 = w.wt3; g.wt3_plane = w.wt3_plane; g.wt_bs = 0; g.bias = w.bias;
    g.x = x; g.y = y; g.epi = epi; g.r = r;
    g.imgs_per_group = 1;
    if (loader == LD_LN) {
      g.mu = mu_; g.rstd = rstd_; g.lnw = nullptr; g.lnb = nullptr;   // folded into w (conv1_ln)
      g.ln_center = with_bias_ln_ ? 1 : 0;
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

  void run_block(const HintBlock& bl, TV x, int B, int H, int W, int mode, size_t& perm_off,
                 hipStream_t s) {
    const int C = bl.C, P = H * W, h = bl.hidden;
    const TV t1 = tv(t1_, (long)t1_per_), t2 = tv(t2_, (long)t2_per_);
    if (bl.att) {
      if (!gemm_ln_inline(C, bl.qkv.M, P, B)) launch_ln_stats(x, mu_, rstd_, B, C, P, s);
      gemm(bl.qkv, x, t1, B, P, EP_STORE, TV{}, s, LD_LN, &bl.n1);
      launch_dwconv3x3(t1, bl.qkv_dw.w, bl.qkv_dw.b, t2, B, 3 * C, H, W, 0, s);
      const HintAttnShape sh = hint_attn_shape(B, C, P, mode);
      HintAttnScratch sc;
      hint_attn_scratch_carve(sh, attn_scratch_, &sc);
      const float* cin = cache_valid_ ? cache_cur_ : nullptr;
      float* cout = (cache_cur_ == cache_a_) ? cache_b_ : cache_a_;
      int* perm = perms_ + perm_off;
      const int* forced = forced_on_ ? forced_ + perm_off : nullptr;
      perm_off += (size_t)sh.G * C;
      launch_hint_attention(sh, tv_chan(t2, 0, P), tv_chan(t2, C, P), bl.aw, cin, cache_c_, cout, sc,
                            perm, forced, s);
      cache_cur_ = cout; cache_c_ = C; cache_g_ = sh.G; cache_valid_ = true;
      // attention * V with the sorted channel order, fused with s = out + q^ + k^
      GemmArgs g{};
    g.splitk_ws = splitk_; g.splitk_floats = SPLITK_WS_FLOATS;
      g.B = B; g.M = C; g.K = C; g.P = P;
      g.Mpad = (C + GEMM_MPAD - 1) / GEMM_MPAD * GEMM_MPAD;
      g.wt = sc.att_t;
      g.wt_bs = (long)((C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD) * g.Mpad;
      // The split kernel for short outputs can take this product too (per-image planes made on the device), but
      // its gathered rows cost a vector multiply-add per load and it runs at 3.1 TB/s against 4.7 for the fp32
      // kernel below (10.6 against 6.9 ms per step at 256 x 256, 32 images): opt-in only, JIRF_ATTN_V_SPLIT=1.
      static const bool attn_v_split = std::getenv("JIRF_ATTN_V_SPLIT") && std::atoi(std::getenv("JIRF_ATTN_V_SPLIT")) != 0;
      if (attn_v_split && gemm_split_bf16_enabled() && C <= 96) {
        launch_split_planes(sc.att_t, sc.att3, B, (C + GEMM_KPAD - 1) / GEMM_KPAD * GEMM_KPAD, g.Mpad, s);
        g.wt3 = sc.att3; g.wt3_plane = g.wt_bs; g.wt3_bs = 3 * g.wt_bs;
      }
      g.x = tv_chan(t2, 2 * C, P);
      g.kidx = perm; g.imgs_per_group = B / sh.G;
      g.y = tv_chan(t1, C, P); g.y2 = tv_chan(t1, 2 * C, P);
      g.epi = EP_AV_INTRA;
      g.q = tv_chan(t2, 0, P); g.k = tv_chan(t2, C, P);
      g.rq = sc.rq; g.rk = sc.rk;
      launch_gemm(g, LD_GATHER, s);
      // intra-cache modulation and project_out: g = gelu(Wg s + bg) * s, then the folded linear tail
      // over [g; out] (rows 0..C-1 and C..2C-1 of t1) with the block's residual
      gemm(bl.gate, tv_chan(t1, 2 * C, P), tv_chan(t1, 0, P), B, P, EP_GELU_MUL, tv_chan(t1, 2 * C, P), s);
      gemm(bl.tail, tv_chan(t1, 0, P), x, B, P, EP_RESID, x, s);
    }
    if (!gemm_ln_inline(C, bl.pin.M, P, B)) launch_ln_stats(x, mu_, rstd_, B, C, P, s);
    gemm(bl.pin, x, t1, B, P, EP_STORE, TV{}, s, LD_LN, &bl.n2);
    launch_dwconv3x3(t1, bl.ffn_dw.w, bl.ffn_dw.b, t2, B, 2 * h, H, W, 1, s);
    gemm(bl.pout, t2, x, B, P, EP_RESID, x, s);
  }

  void run(const float* x, float* y, int B, int H, int W, int mode, hipStream_t s) {
    const int d = dim_;
    const long P1 = (long)H * W, P2 = P1 / 4, P3 = P1 / 16, P4 = P1 / 64;
    const int H2 = H / 2, W2 = W / 2, H3 = H / 4, W3 = W / 4, H4 = H / 8, W4 = W / 8;
    const TV xin = tv(const_cast<float*>(x), (long)inp_ch_ * P1);
    const TV yout = tv(y, (long)out_ch_ * P1);
    const TV cat1 = tv(cat1_, 2L * d * P1), cat2 = tv(cat2_, 4L * d * P2), cat3 = tv(cat3_, 8L * d * P3);
    const TV x2 = tv(x2_, 2L * d * P2), x3 = tv(x3_, 4L * d * P3), x4 = tv(x4_, 8L * d * P4);
    const TV e1 = tv_chan(cat1, d, P1), e2 = tv_chan(cat2, 2 * d, P2), e3 = tv_chan(cat3, 4 * d, P3);

    // the carried cache lives in cache_store_ between calls; bring it into the ping-pong pair
    if (cache_valid_) {
      JIRF_REQUIRE(cache_g_ == (mode == JIRF_BATCH_COUPLED ? 1 : B), JIRF_ESTATE,
                   "qv_cache was produced with another batch layout; call reset_cache");
      cache_cur_ = static_cast<float*>(cache_store_.p);
    } else {
      cache_cur_ = nullptr;
    }
    size_t perm_off = 0;
    auto run_stage = [&](int idx, TV t, int h, int w) {
      for (auto& bl : stages_[idx].blocks) run_block(bl, t, B, h, w, mode, perm_off, s);
    };

    conv3(patch_, xin, e1, B, H, W, EP_STORE, TV{}, s);
    if (dual_pixel_)
      JIRF_HIP_CHECK(hipMemcpy2DAsync(emb_, (size_t)d * P1 * 4, e1.p, (size_t)e1.bs * 4, (size_t)d * P1 * 4, B,
                                      hipMemcpyDeviceToDevice, s));
    run_stage(0, e1, H, W);
    conv3(down12_, e1, e2, B, H, W, EP_UNSHUFFLE, TV{}, s);
    run_stage(1, e2, H2, W2);
    conv3(down23_, e2, e3, B, H2, W2, EP_UNSHUFFLE, TV{}, s);
    run_stage(2, e3, H3, W3);
    conv3(down34_, e3, x4, B, H3, W3, EP_UNSHUFFLE, TV{}, s);
    run_stage(3, x4, H4, W4);

    conv3(up43_, x4, cat3, B, H4, W4, EP_SHUFFLE, TV{}, s);
    gemm(red3_, cat3, x3, B, (int)P3, EP_STORE, TV{}, s);
    run_stage(4, x3, H3, W3);

    conv3(up32_, x3, cat2, B, H3, W3, EP_SHUFFLE, TV{}, s);
    gemm(red2_, cat2, x2, B, (int)P2, EP_STORE, TV{}, s);
    run_stage(5, x2, H2, W2);

    conv3(up21_, x2, cat1, B, H2, W2, EP_SHUFFLE, TV{}, s);
    run_stage(6, cat1, H, W);
    run_stage(7, cat1, H, W);

    if (dual_pixel_) {
      gemm(skip_, tv(emb_, (long)d * P1), cat1, B, (int)P1, EP_RESID, cat1, s);
      conv3(out_, cat1, yout, B, H, W, EP_STORE, TV{}, s);
    } else {
      conv3(out_, cat1, yout, B, H, W, EP_RESID, xin, s);
    }

    // keep the cache beyond the arena's ping-pong pair (the next call may re-plan)
    if (cache_valid_ && cache_cur_ != cache_store_.p) {
      const size_t bytes = (size_t)cache_g_ * cache_c_ * cache_c_ * 4;
      JIRF_REQUIRE(cache_store_.bytes >= bytes, JIRF_ESTATE, "cache store not sized");
      JIRF_HIP_CHECK(hipMemcpyAsync(cache_store_.p, cache_cur_, bytes, hipMemcpyDeviceToDevice, s));
    }
  }

  // called by plan(); keeps the persistent cache store large enough for any block's C x C
  void ensure_cache_store(int G) {
    const size_t need = (size_t)G * cmax_ * cmax_ * 4;
    if (cache_store_.bytes < need) {
      // growing loses the contents: preserve a valid cache across the reallocation
      if (cache_valid_) {
        DeviceBuffer tmp;
        tmp.ensure(need);
        JIRF_HIP_CHECK(hipDeviceSynchronize());
        JIRF_HIP_CHECK(hipMemcpy(tmp.p, cache_store_.p, (size_t)cache_g_ * cache_c_ * cache_c_ * 4,
                                 hipMemcpyDeviceToDevice));
        std::swap(tmp.p, cache_store_.p);
        std::swap(tmp.bytes, cache_store_.bytes);
      } else {
        cache_store_.ensure(need);
      }
    }
  }

 private:
  int inp_ch_, out_ch_, dim_, refine_;
  std::vector<int> nb_;
  double ffn_;
  bool bias_, with_bias_ln_, dual_pixel_;
  std::vector<Stage> stages_;
  WeightPool pool_;
  PackedW patch_, down12_, down23_, down34_, up43_, up32_, up21_, red3_, red2_, skip_, out_;

  float* splitk_ = nullptr;
  DeviceBuffer arena_, cache_store_;
  int pB_ = 0, pH_ = 0, pW_ = 0, pmode_ = -1;
  float *in_ = nullptr, *outb_ = nullptr, *cat1_ = nullptr, *cat2_ = nullptr, *cat3_ = nullptr;
  float *x2_ = nullptr, *x3_ = nullptr, *x4_ = nullptr, *emb_ = nullptr, *t1_ = nullptr, *t2_ = nullptr;
  float *mu_ = nullptr, *rstd_ = nullptr, *cache_a_ = nullptr, *cache_b_ = nullptr;
  size_t t1_per_ = 0, t2_per_ = 0, perm_total_ = 0;
  void* attn_scratch_ = nullptr;
  int *perms_ = nullptr, *forced_ = nullptr;
  bool forced_on_ = false;
  int cmax_ = 0;
  // cache state
  bool cache_valid_ = false;
  float* cache_cur_ = nullptr;
  int cache_c_ = 0, cache_g_ = 0;

};

}  // namespace

std::unique_ptr<Net> make_hint_net(const Config& cfg) { return std::make_unique<HintNet>(cfg); }

}  // namespace jirf
