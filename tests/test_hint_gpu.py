"""HINT parity on the GPU: attention core (kernel level) and the whole network through the
drop-in class, against the oracle.

The channel order of every attention block is an argsort of correlation means that are often
1e-7 apart, so it is discontinuous in its input: two correct fp32 implementations can disagree
on a near tie, and one swapped pair changes everything downstream.  The whole-network tests
therefore (a) check the free-running orders block by block, tolerating a disagreement only where
the oracle's own margin is below RANK_MARGIN, and (b) assert the 1e-3 output bound with the
orders pinned to the oracle's (jirf_net_set_forced_perms).
"""
import numpy as np
import pytest
import torch

from oracle import hint_oracle as ho
from oracle.common import psnr
from oracle.weights import make_hint_state, synthetic_images
from tests.gpu_util import call, dev, hptr, max_abs, ptr, stream

pytestmark = pytest.mark.gpu

TOL = 1e-3            # north-star bound on fp32 outputs (max-abs)
RANK_MARGIN = 1e-6    # below this gap between adjacent similarities a rank flip is not a defect


def _core_params(C, rng, mod_gain=1.0):
    p, flat = {}, []
    t = rng.uniform(0.5, 2.0, (4, 1, 1)).astype(np.float32)
    p["a.temperature"] = torch.from_numpy(t)
    flat.append(t.reshape(-1))
    for g, cg in enumerate(ho.group_sizes(C)):
        for name, shape, scale in ((".conv_width.weight", (2 * cg, cg, 1), mod_gain / np.sqrt(cg)),
                                   (".conv_width.bias", (2 * cg,), 0.05 * mod_gain),
                                   (".gating_conv.weight", (cg, cg, 1), mod_gain / np.sqrt(cg)),
                                   (".gating_conv.bias", (cg,), 0.05 * mod_gain)):
            a = (rng.standard_normal(shape) * scale).astype(np.float32)
            p[f"a.inter_modulator{g + 1}{name}"] = torch.from_numpy(a)
            flat.append(a.reshape(-1))
    return p, np.concatenate(flat)


def _min_gap(sim):
    s = torch.sort(sim, descending=True).values
    return float((s[:-1] - s[1:]).min())


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,C,P,cache_c,mod_gain", [
    (1, 96, 48 * 64, 0, 1.0),       # cold start
    (2, 96, 1000, 96, 1.0),         # same-size cache, ragged pixel count
    (1, 192, 24 * 32, 384, 1.0),    # cache pooled down (latent -> decoder level 3)
    (2, 384, 12 * 16, 96, 1.0),     # cache pooled up (refinement -> latent of the next call)
    (1, 96, 4096, 96, 30.0),        # strong modulation: drives the floor-division off zero
])
def test_attention_core(mode, B, C, P, cache_c, mod_gain):
    rng = np.random.default_rng(100 + C + P)
    qkv = rng.standard_normal((B, 3 * C, P)).astype(np.float32)
    # a shared component with a distinct loading per channel separates the similarity means
    # (adjacent means ~1e-3 apart), so the expected channel order is well defined
    load = rng.permutation(np.linspace(0.2, 1.5, C)).astype(np.float32)
    common = rng.standard_normal((1, 1, P)).astype(np.float32)
    qkv[:, :C] += load[None, :, None] * common
    G = 1 if mode == 1 else B
    cache = None
    if cache_c:
        cache = (rng.standard_normal((G, cache_c, cache_c)) * 0.5).astype(np.float32)
    p, flat = _core_params(C, rng, mod_gain)
    sw = ho.HintSwitches()
    q, k, v = [torch.from_numpy(qkv[:, i * C:(i + 1) * C]) for i in range(3)]

    out = torch.empty(B, C, P, device="cuda")
    s_out = torch.empty(B, C, P, device="cuda")
    perm = torch.empty(G, C, dtype=torch.int32, device="cuda")
    att = torch.zeros(B, C, C, device="cuda")
    cache_out = torch.empty(G, C, C, device="cuda")
    qd = dev(qkv)
    cd = None if cache is None else dev(cache)
    call("jirf_op_hint_attention_core", ptr(qd), hptr(flat), ptr(cd), cache_c, ptr(out), ptr(s_out), ptr(perm),
         ptr(att), ptr(cache_out), B, C, P, mode, stream())

    sizes = ho.group_sizes(C)
    for g in range(G):
        sl = slice(g, g + 1) if mode == 0 else slice(0, B)
        cg = None if cache is None else torch.from_numpy(cache[g:g + 1])
        taps = {}
        ro, rt, ratt, rcache, order = ho.hmha_core(p, "a", q[sl], k[sl], v[sl], cg, sw,
                                                   tap=lambda n, t: taps.__setitem__(n, t))
        gap = _min_gap(taps["a.mean_similarity"])
        got = perm[g].cpu().numpy()
        assert sorted(got.tolist()) == list(range(C)), "device channel order is not a permutation"
        if gap > RANK_MARGIN:
            assert np.array_equal(got, order.numpy()), f"channel order differs (oracle min gap {gap:.2e})"
        elif not np.array_equal(got, order.numpy()):
            # a near tie in the oracle's similarity means: the order itself is undefined there, but every
            # other output is a function of the order, so evaluate the oracle on the device's order and
            # keep asserting (the swapped channels must be the near-tied ones)
            sim = taps["a.mean_similarity"].numpy()
            moved = np.nonzero(got != order.numpy())[0]
            span = np.abs(sim[got[moved]] - sim[order.numpy()[moved]]).max()
            assert span < 10 * RANK_MARGIN, f"order differs beyond a near tie (similarity span {span:.2e})"
            print(f"[attention_core C={C} P={P}] near tie (gap {gap:.2e}): oracle re-evaluated on the device order")
            ro, rt, ratt, rcache, _ = ho.hmha_core(p, "a", q[sl], k[sl], v[sl], cg, sw,
                                                   forced_order=torch.from_numpy(got.astype(np.int64)))
        assert max_abs(out[sl].cpu(), ro) < 2e-5 * max(1.0, float(ro.abs().max()))
        assert max_abs(s_out[sl].cpu(), ro + rt) < 2e-5 * max(1.0, float(ro.abs().max()))
        a_gpu = att[sl].cpu()
        s0 = 0
        for gi, c in enumerate(sizes):
            assert max_abs(a_gpu[:, s0:s0 + c, s0:s0 + c], ratt[gi]) < 1e-5 * max(1.0, float(ratt[gi].abs().max()))
            s0 += c
        # the cache holds floor()ed values: compare exactly where the oracle's pre-floor value is
        # not within 1e-4 of an integer, and report how many entries were decided near a boundary
        diff = (cache_out[g].cpu() - rcache[0]).abs()
        assert float(diff.max()) < 0.11, "cache differs by more than one floor step"
        frac_off = float((diff > 1e-5).float().mean())
        assert frac_off < 1e-3, f"{frac_off:.2%} of cache entries differ"
        if mod_gain > 1:
            assert float(rcache.abs().max()) > 0.05, "test did not drive the cache off zero"


def _build(cfg, sd, batch_mode):
    from basicsr.models.archs import define_network
    net = define_network(dict(cfg, type="HINT"))
    net.load_state_dict(sd)
    net.cuda().eval()
    net.batch_mode = batch_mode
    return net


def _oracle_run(sd, x, cfg, cache=None, forced=None):
    taps = {}
    y, c = ho.hint_forward(sd, torch.from_numpy(x), cfg, cache, tap=lambda n, t: taps.__setitem__(n, t),
                           forced_orders=forced)
    names = [n[:-len(".attn.perm")] for n in taps if n.endswith(".attn.perm")]
    perms = [taps[n + ".attn.perm"].numpy().astype(np.int32) for n in names]
    gaps = [_min_gap(taps[n + ".attn.mean_similarity"]) for n in names]
    return y.numpy(), c.numpy(), names, perms, gaps


SMALL_CFG = dict(inp_channels=3, out_channels=3, dim=16, num_blocks=[1, 2, 2, 3], num_refinement_blocks=1,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, ln_type="WithBias",
                 dual_pixel_task=False)


@pytest.mark.parametrize("cfg_name,H,W", [("small", 64, 96), ("small_biasfree", 40, 72), ("full", 64, 64),
                                          ("small_dim24", 32, 48)])
def test_network_b1_cold_and_warm(cfg_name, H, W):
    cfg = ho.default_config() if cfg_name == "full" else dict(SMALL_CFG)
    if cfg_name == "small_biasfree":
        cfg.update(ln_type="BiasFree", bias=True)
    if cfg_name == "small_dim24":       # channel groups of 6 / 12 / 18 and 9 / 18 / 27 ...: k-step counts off the batch size
        cfg.update(dim=24)
    sd = make_hint_state(cfg, seed=3, gain=0.6)
    x1 = synthetic_images(1, H, W, seed=11, sigma=50)
    x2 = synthetic_images(1, H, W, seed=12, sigma=50)
    net = _build(cfg, sd, "coupled")

    # ---- free-running: the orders must agree wherever the oracle's margin is healthy ----
    y_ref, c_ref, names, perms, gaps = _oracle_run(sd, x1, cfg)
    y = net(torch.from_numpy(x1).cuda()).cpu().numpy()
    got = net.last_perms(1)
    first_bad = None
    for i, n in enumerate(names):
        if not np.array_equal(got[i][0], perms[i]):
            first_bad = i
            break
        # everything before the first disagreement ran on identical orders
    print(f"[{cfg_name}] free-running: first differing block = "
          f"{None if first_bad is None else names[first_bad]}, min oracle gap = {min(gaps):.2e}, "
          f"max|dy| = {max_abs(y, y_ref):.2e}")
    if first_bad is None:
        assert max_abs(y, y_ref) < TOL
        assert abs(psnr(y, x1) - psnr(y_ref, x1)) < 0.01
    else:
        assert gaps[first_bad] < RANK_MARGIN, (
            f"order of {names[first_bad]} differs although the oracle's gap is {gaps[first_bad]:.2e}")

    # ---- pinned orders: cold call, then a warm call that continues the carried cache ----
    net.qv_cache = None
    net.force_perms([p[None] for p in perms])
    y = net(torch.from_numpy(x1).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL
    assert abs(psnr(y, x1) - psnr(y_ref, x1)) < 0.01
    cache = net.qv_cache
    assert cache.shape == c_ref.shape
    assert float(np.mean(np.abs(cache - c_ref) > 1e-5)) < 1e-3

    y2_ref, c2_ref, _, perms2, _ = _oracle_run(sd, x2, cfg, cache=torch.from_numpy(c_ref))
    net.qv_cache = c_ref
    net.force_perms([p[None] for p in perms2])
    y2 = net(torch.from_numpy(x2).cuda()).cpu().numpy()
    assert max_abs(y2, y2_ref) < TOL
    assert float(np.mean(np.abs(net.qv_cache - c2_ref) > 1e-5)) < 1e-3


@pytest.mark.parametrize("batch_mode", ["coupled", "per_image"])
def test_network_batch_semantics(batch_mode):
    cfg = dict(SMALL_CFG)
    sd = make_hint_state(cfg, seed=5, gain=0.6)
    B, H, W = 3, 48, 64
    x = synthetic_images(B, H, W, seed=21, sigma=50)
    net = _build(cfg, sd, batch_mode)
    if batch_mode == "coupled":
        y_ref, c_ref, names, perms, gaps = _oracle_run(sd, x, cfg)
        forced = [p[None] for p in perms]
    else:
        runs = [_oracle_run(sd, x[i:i + 1], cfg) for i in range(B)]
        y_ref = np.concatenate([r[0] for r in runs])
        c_ref = np.concatenate([r[1] for r in runs])
        forced = [np.stack([r[3][k] for r in runs]) for k in range(len(runs[0][3]))]
    net.native().plan(B, H, W, 1 if batch_mode == "coupled" else 0)
    net.force_perms(forced)
    y = net(torch.from_numpy(x).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL
    cache = net.qv_cache
    assert cache.shape == c_ref.shape
    assert float(np.mean(np.abs(cache - c_ref) > 1e-5)) < 1e-3
    # free-running orders of the same call (diagnostic: count disagreements with healthy margins)
    net.qv_cache = None
    net.force_perms(None)
    net(torch.from_numpy(x).cuda())


def test_host_pointer_path_matches_device_path():
    cfg = dict(SMALL_CFG)
    sd = make_hint_state(cfg, seed=9, gain=0.6)
    x = synthetic_images(1, 32, 40, seed=2, sigma=50)
    net = _build(cfg, sd, "coupled")
    y_dev = net(torch.from_numpy(x).cuda()).cpu().numpy()
    net.qv_cache = None
    y_host = net(x)                       # numpy in, numpy out: copies through the staging buffers
    assert isinstance(y_host, np.ndarray)
    assert np.array_equal(y_dev, y_host)  # same kernels, same order: bit-identical


def test_rejects_bad_shapes():
    cfg = dict(SMALL_CFG)
    net = _build(cfg, make_hint_state(cfg, seed=1), "coupled")
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 30, 32).cuda())


def test_network_720p_against_oracle():
    """1280 x 720 (the motion-deblurring frame size): the long H*W contraction of the channel
    attention and every index computation at 0.9 Mpx, small config so the oracle finishes in seconds."""
    cfg = dict(SMALL_CFG)
    sd = make_hint_state(cfg, seed=4, gain=0.6)
    x = synthetic_images(1, 720, 1280, seed=31, sigma=25)
    y_ref, c_ref, names, perms, gaps = _oracle_run(sd, x, cfg)
    net = _build(cfg, sd, "per_image")
    net.native().plan(1, 720, 1280, 0)
    net.force_perms([p[None] for p in perms])
    y = net(torch.from_numpy(x).cuda()).cpu().numpy()
    assert max_abs(y, y_ref) < TOL
    assert abs(psnr(y, x) - psnr(y_ref, x)) < 0.01
    assert float(np.mean(np.abs(net.qv_cache - c_ref) > 1e-5)) < 1e-3


def test_full_config_720p_batch_consistency():
    """Full HINT at 1280 x 720: a frame restored alone and as one image of a per-image batch give
    bit-identical results (same kernels, same split-K schedule per image), and reruns repeat."""
    cfg = ho.default_config()
    sd = make_hint_state(cfg, seed=6, gain=0.6)
    x = synthetic_images(2, 720, 1280, seed=41, sigma=50)
    net = _build(cfg, sd, "per_image")
    xd = torch.from_numpy(x).cuda()
    y2 = net(xd).cpu().numpy()
    perms2 = net.last_perms(2)
    assert np.isfinite(y2).all()
    net.qv_cache = None
    y2b = net(xd).cpu().numpy()
    assert np.array_equal(y2, y2b)
    net.qv_cache = None
    net.native().plan(1, 720, 1280, 0)
    net.force_perms([p[1:2] for p in perms2])
    y1 = net(xd[1:2]).cpu().numpy()
    # B = 1 and B = 2 choose different split-K counts for the statistics, so sums differ in the
    # last bits; with the channel orders pinned the outputs agree far inside the parity bound
    assert max_abs(y1, y2[1:2]) < 1e-4


def test_network_dual_pixel_task():
    """dual_pixel_task=True (hint_arch.py:700-703): six input channels, the 1x1 skip from the patch
    embedding, no global residual."""
    cfg = dict(SMALL_CFG, dual_pixel_task=True, inp_channels=6, bias=True)
    sd = make_hint_state(cfg, seed=8, gain=0.6)
    x = np.concatenate([synthetic_images(1, 48, 64, seed=61, sigma=25), synthetic_images(1, 48, 64, seed=62)], axis=1)
    y_ref, c_ref, names, perms, gaps = _oracle_run(sd, x, cfg)
    net = _build(cfg, sd, "coupled")
    net.native().plan(1, 48, 64, 1)
    net.force_perms([p[None] for p in perms])
    y = net(torch.from_numpy(x).cuda()).cpu().numpy()
    assert y.shape == (1, 3, 48, 64)
    assert max_abs(y, y_ref) < TOL
