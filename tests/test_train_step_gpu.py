"""The AST optimisation step (reference basicsr/models/image_restoration_model.py:235-276) on the GPU against
torch autograd over the oracle's forward (oracle/ast_oracle.py, torch-CPU fp32; "parity unpinned": Jittor's
own autograd cannot run here) and a NumPy restatement of Jittor's clip_grad_norm + Adam:

 * every parameter's gradient of the smooth-L1 loss, full depth-2 AST on a batch of two 128 x 128 crops
   (the literal batched call: the window / head pairing of K runs over the whole batch), within 1e-3 of
   autograd's, and the fraction of parameters covered (all of them);
 * the weights after iteration 1 (no clip) and iteration 2 (clip 0.01) against the oracle's;
 * the training forward equals the inference forward of the same weights (stochastic depth off);
 * stochastic depth (DropPath): the per-block rates, the gradients under given per-sample factors against
   autograd over the oracle with the same factors, and the statistics of the drawn factors.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ast_oracle as so
from oracle.weights import make_ast_state
from tests.gpu_util import max_abs

pytestmark = pytest.mark.gpu
REL = 1e-3


def _cfg():
    cfg = so.default_config()
    # shifted blocks at the bottleneck and at full resolution; as many decoder blocks as encoder blocks (the
    # reference's drop-path schedule indexes the reversed encoder list by decoder block, ast_arch.py:983-1190)
    cfg["depths"] = [2, 1, 1, 1, 2, 1, 1, 1, 2]
    return cfg


def _state(cfg, seed):
    """Synthetic AST weights, trainable tensors only (the index buffer of the bias table is not a parameter)."""
    return {k: np.asarray(v, np.float32) for k, v in make_ast_state(cfg, seed=seed, gain=0.5).items()
            if not k.endswith("relative_position_index")}


def _data(B, S, seed):
    rng = np.random.default_rng(seed)
    gt = rng.random((B, 3, S, S), dtype=np.float32)
    lq = np.clip(gt + 0.1 * rng.standard_normal((B, 3, S, S)).astype(np.float32), 0, 1).astype(np.float32)
    return lq, gt


def _oracle_grads(sd, lq, gt, cfg, drop_scales=None):
    params = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in sd.items()}
    pred = so.ast_forward(params, torch.from_numpy(lq), cfg, drop_scales=drop_scales)
    loss = F.smooth_l1_loss(pred, torch.from_numpy(gt), reduction="none").mean()      # losses.py:12-14
    loss.backward()
    return float(loss.detach()), pred.detach().numpy(), {k: v.grad.numpy() for k, v in params.items()}


def _jittor_adam(p, g, m, v, step, lr=2e-4, b0=0.9, b1=0.999, eps=1e-8, clip=None):
    """jt.optim.Optimizer.clip_grad_norm(max_norm, 2) over all parameters, then jt.optim.Adam.step."""
    if clip is not None:
        total = np.sqrt(sum(float(np.sum(x.astype(np.float64) ** 2)) for x in g.values()))
        coef = min(clip / (total + 1e-6), 1.0)
        g = {k: x * np.float32(coef) for k, x in g.items()}
    out = {}
    for k in p:
        m[k] = b0 * m[k] + (1 - b0) * g[k]
        v[k] = b1 * v[k] + (1 - b1) * g[k] * g[k]
        out[k] = p[k] - m[k] * (lr * np.sqrt(1 - b1 ** step) / (1 - b0 ** step)) / (np.sqrt(v[k]) + eps)
    return out


def test_gradients_and_two_steps_against_the_oracle():
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 11)
    lq, gt = _data(2, 128, 3)
    loss_ref, pred_ref, g_ref = _oracle_grads(sd, lq, gt, cfg)

    tr = AstTrainStep(cfg, sd, lr=2e-4, coupled=True, drop_path_rate=0.0)
    loss, pred = tr.forward_backward(torch.from_numpy(lq), torch.from_numpy(gt))
    assert max_abs(pred.cpu().numpy(), pred_ref) < 1e-3
    assert abs(loss - loss_ref) < 1e-5 * abs(loss_ref)
    g = tr.grad_dict()
    worst, covered = ("", 0.0), 0
    for k, want in g_ref.items():
        scale = max(float(np.abs(want).max()), 1e-12)
        e = float(np.abs(g[k] - want).max()) / scale
        if e > worst[1]:
            worst = (k, e)
        covered += want.size if e < REL else 0
        assert e < REL, f"{k}: relative error {e:.2e}"
    total = sum(v.size for v in g_ref.values())
    print(f"[AST training step] loss {loss:.6f} (oracle {loss_ref:.6f}); gradients verified for {covered}/{total} "
          f"parameters ({len(g_ref)} tensors); worst {worst[0]} {worst[1]:.2e}")
    assert covered == total == tr.num_parameters()

    # ---- iteration 1 (clip only from the second iteration on, image_restoration_model.py:269), then iteration 2
    m = {k: np.zeros_like(v) for k, v in sd.items()}
    v = {k: np.zeros_like(x) for k, x in sd.items()}
    p1_ref = _jittor_adam(sd, g_ref, m, v, 1)
    tr.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), current_iter=1)
    p1 = tr.state_dict()
    for k in sd:
        # the first Adam step moves every weight by ~lr whatever the gradient's size (m / sqrt(v) = +-1), so the
        # comparison is of the step direction: a wrong sign would show as 2 lr
        assert np.abs(p1[k] - p1_ref[k]).max() < 0.05 * 2e-4 + 1e-6 * np.abs(sd[k]).max(), k
    lq2, gt2 = _data(2, 128, 4)
    p1f = {k: np.asarray(x, np.float32) for k, x in p1_ref.items()}
    _, _, g2_ref = _oracle_grads(p1f, lq2, gt2, cfg)
    p2_ref = _jittor_adam(p1f, g2_ref, m, v, 2, clip=0.01)
    loss2 = tr.optimize_parameters(torch.from_numpy(lq2), torch.from_numpy(gt2), current_iter=2)
    p2 = tr.state_dict()
    err = max(float(np.abs(p2[k] - p2_ref[k]).max()) for k in sd)
    moved = max(float(np.abs(p2_ref[k] - sd[k]).max()) for k in sd)
    print(f"[AST training step] after two iterations: max |w - w_oracle| = {err:.2e} (weights moved by up to {moved:.2e}); "
          f"loss {loss2:.6f}")
    assert err < 0.1 * moved


def test_gradients_at_the_benchmark_configuration():
    """The step `bench.py --train-step` times: full AST (constructor defaults, depths [2] * 9, token_mlp frfn,
    26.3 M parameters) on two 256 x 256 crops, the literal batched call.  Loss and every parameter's gradient
    against autograd over the oracle."""
    from jirf_hip.train import AstTrainStep
    cfg = so.default_config()
    sd = _state(cfg, 1)
    lq, gt = _data(2, 256, 13)
    loss_ref, pred_ref, g_ref = _oracle_grads(sd, lq, gt, cfg)
    tr = AstTrainStep(cfg, sd, coupled=True, drop_path_rate=0.0)
    loss, pred = tr.forward_backward(torch.from_numpy(lq), torch.from_numpy(gt))
    assert max_abs(pred.cpu().numpy(), pred_ref) < 1e-3
    assert abs(loss - loss_ref) < 1e-5 * abs(loss_ref)
    g = tr.grad_dict()
    worst = max(((float(np.abs(g[k] - w).max()) / max(float(np.abs(w).max()), 1e-12), k) for k, w in g_ref.items()))
    print(f"[AST training step, full config, 2 x 256 x 256] loss {loss:.6f} (oracle {loss_ref:.6f}); "
          f"{tr.num_parameters()} parameters, worst relative gradient error {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < REL
    assert sum(w.size for w in g_ref.values()) == tr.num_parameters() == 26298387


def test_training_forward_equals_the_inference_forward():
    from basicsr.models.archs import define_network
    from jirf_hip.train import AstTrainStep
    cfg = so.default_config()
    sd = _state(cfg, 5)
    lq, gt = _data(2, 128, 7)
    net = define_network(dict(cfg, type="AST"))
    net.load_state_dict(sd)
    net.cuda().eval()
    for coupled, mode in ((False, "per_image"), (True, "coupled")):
        tr = AstTrainStep(cfg, sd, coupled=coupled, drop_path_rate=0.0)
        _, pred = tr.forward_backward(torch.from_numpy(lq), torch.from_numpy(gt), backward=False)
        net.batch_mode = mode
        y = net(torch.from_numpy(lq).cuda())
        assert max_abs(pred.cpu().numpy(), y.cpu().numpy()) < 2e-5, mode


def test_stochastic_depth_against_the_oracle_with_the_same_factors():
    """DropPath in training mode (reference ast_arch.py:52-70, :837, :840, rates :981-983).  The reference draws
    the per-sample keep decisions from Jittor's generator; here they are given, to the GPU step and to the oracle
    alike, at a rate (0.5) that drops many branches: per-block rates, prediction, loss and every gradient."""
    from jirf_hip.train import AstTrainStep
    cfg = dict(_cfg(), drop_path_rate=0.5)
    sd = _state(cfg, 17)
    lq, gt = _data(2, 128, 19)
    tr = AstTrainStep(cfg, sd, coupled=True)
    rates = {b["name"]: b["drop"] for b in tr._blocks()}
    want_rates = so.drop_path_rates(cfg)
    assert rates.keys() == want_rates.keys() and all(abs(rates[k] - want_rates[k]) < 1e-7 for k in rates)
    assert rates["encoderlayer_0.blocks.0"] == 0.0 and rates["conv.blocks.1"] == 0.5 and rates["decoderlayer_0.blocks.0"] == 0.5
    rng = np.random.default_rng(23)
    scales = {}
    for name, r in want_rates.items():
        for branch in (("attn", "mlp") if not name.startswith(("encoderlayer", "dowsample")) else ("mlp",)):
            if r > 0:
                scales[f"{name}.{branch}"] = ((rng.random(2) < 1 - r) / (1 - r)).astype(np.float32)
    dropped = sum(int((v == 0).sum()) for v in scales.values())
    assert dropped >= 3                                       # the case exercises dropped branches
    loss_ref, pred_ref, g_ref = _oracle_grads(sd, lq, gt, cfg, drop_scales=scales)
    tr.drop_scales = scales
    loss, pred = tr.forward_backward(torch.from_numpy(lq), torch.from_numpy(gt))
    assert max_abs(pred.cpu().numpy(), pred_ref) < 1e-3
    assert abs(loss - loss_ref) < 1e-5 * abs(loss_ref)
    g = tr.grad_dict()
    worst = max(((float(np.abs(g[k] - w).max()) / max(float(np.abs(w).max()), 1e-12), k) for k, w in g_ref.items()))
    print(f"[AST training step, DropPath] {dropped} of {sum(v.size for v in scales.values())} branch samples dropped; "
          f"worst relative gradient error {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < REL
    # and it differs from the step without stochastic depth (the factors are applied)
    loss0, _, _ = _oracle_grads(sd, lq, gt, cfg)
    assert abs(loss0 - loss_ref) > 1e-4 * abs(loss_ref)


def test_stochastic_depth_draws():
    """Drawn factors: 0 or 1 / keep_prob per sample, about keep_prob of them kept, the same stream for the same
    seed, none drawn at rate 0, and evaluation (`backward=False` on the inference class) is unaffected."""
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 3)
    a = AstTrainStep(cfg, sd, seed=5)                        # rate 0.1: the constructor default of the reference
    assert a.drop_path_rate == pytest.approx(0.1)
    kept = total = 0
    for _ in range(40):
        a._draw_scales(8)
        for key, f in a._scales.items():
            r = {b["name"]: b["drop"] for b in a._blocks()}[key.rsplit(".", 1)[0]]
            vals = f.cpu().numpy()
            assert np.all((vals == 0) | (np.abs(vals - 1 / (1 - r)) < 1e-6))
            if abs(r - 0.1) < 1e-9:
                kept += int((vals > 0).sum())
                total += vals.size
    assert total > 500 and abs(kept / total - 0.9) < 0.05
    b, c = AstTrainStep(cfg, sd, seed=5), AstTrainStep(cfg, sd, seed=6)
    a2 = AstTrainStep(cfg, sd, seed=5)
    a2._draw_scales(8); b._draw_scales(8); c._draw_scales(8)
    assert all(torch.equal(a2._scales[k], b._scales[k]) for k in b._scales)
    assert any(not torch.equal(c._scales[k], b._scales[k]) for k in b._scales)
    z = AstTrainStep(cfg, sd, drop_path_rate=0.0)
    z._draw_scales(8)
    assert z._scales == {}


def test_captured_step_equals_the_step_queued_kernel_by_kernel():
    """use_graph=True replays forward + backward as one hipGraph from the second iteration on: same kernels on the
    same data in the same order, so weights and losses are bit-identical to the eager trainer's, with stochastic
    depth on (the factors are drawn on the device from the seed, outside the captured region) and across a change
    of input shape (a new capture)."""
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 6)
    eager = AstTrainStep(cfg, sd, seed=3)
    graph = AstTrainStep(cfg, sd, seed=3, use_graph=True)
    assert eager.drop_path_rate == graph.drop_path_rate == pytest.approx(0.1)
    shapes = [(2, 128)] * 4 + [(1, 128)] * 2 + [(2, 128)]
    for it, (B, S) in enumerate(shapes, start=1):
        lq, gt = _data(B, S, 50 + it)
        a = eager.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), it)
        b = graph.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), it)
        assert a == b, (it, a, b)
        assert torch.equal(eager.output, graph.output)
    assert graph._graph is not None
    assert torch.equal(eager.p, graph.p) and torch.equal(eager.m, graph.m) and torch.equal(eager.v, graph.v)


_GROWING_SHAPES_SCRIPT = r"""
import sys, numpy as np, torch
sys.path[:0] = [sys.argv[1], sys.argv[1] + "/image-restoration_amd"]
from tests.test_train_step_gpu import _cfg, _state, _data
from jirf_hip import lib
from jirf_hip.train import AstTrainStep
cfg = _cfg(); sd = _state(cfg, 6)
shapes = [(1, 128)] * 3 + [(2, 128)] * 3 + [(2, 256)] * 2 + [(2, 128)] * 2
def run(tr, other=None):
    out = []
    for it, (B, S) in enumerate(shapes, start=1):
        lq, gt = _data(B, S, 70 + it)
        out.append((tr.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), it), tr.output.clone()))
        if other is not None and it == 6:
            # a second trainer of the process grows the shared scratch while `tr` holds a captured step
            g0 = lib().jirf_train_scratch_generation()
            lq2, gt2 = _data(2, 256, 5)
            other.forward_backward(torch.from_numpy(lq2), torch.from_numpy(gt2))
            assert lib().jirf_train_scratch_generation() > g0, "the larger shape was expected to grow the scratch"
    return out
# the captured trainer runs FIRST and alone: nothing has sized the process-wide scratch for it
graph = AstTrainStep(cfg, sd, seed=3, use_graph=True)
a = run(graph, AstTrainStep(cfg, sd, seed=9))
assert graph._graph is not None
# a call that would have to grow the scratch inside a capture is refused with a clear error, leaves the capture and
# the buffers intact, and works once an eager call has sized the scratch
import ctypes as C
import jirf_hip
M, K, P = 8192, 4096, 64 * 64          # 134 MB of partial sums: more than any AST step of this file needs
dy = torch.randn(1, M, P, device="cuda"); x = torch.randn(1, K, P, device="cuda")
dw = torch.zeros(M, K, device="cuda")
def bwd_weight():
    return lib().jirf_train_conv_bwd_weight(dy.data_ptr(), 0, x.data_ptr(), 0, dw.data_ptr(), None, 1, M, K, 64, 64, 64, 64,
                                            1, 1, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
g0 = lib().jirf_train_scratch_generation()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    rc = bwd_weight()
assert rc == -3 and "stream capture" in jirf_hip.last_error(), (rc, jirf_hip.last_error())
assert lib().jirf_train_scratch_generation() == g0
assert bwd_weight() == 0 and lib().jirf_train_scratch_generation() > g0
torch.cuda.synchronize()
want = dw.clone(); dw.zero_()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    rc = bwd_weight()
assert rc == 0
gr.replay(); torch.cuda.synchronize()
assert torch.equal(dw, want)
# the trainer's fallback: a capture attempted at a shape no eager step has run at either succeeds (nothing had to
# grow) or is refused and the step runs kernel by kernel; both give a finite loss and the next step is captured
fresh = AstTrainStep(cfg, sd, seed=3, use_graph=True)
lq, gt = _data(1, 128, 71)
fresh.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), 1)
fresh._eager_shapes.add((2, 3, 384, 384))
lq, gt = _data(2, 384, 72)
l2 = fresh.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), 2)
l3 = fresh.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), 3)
assert np.isfinite(l2) and np.isfinite(l3) and fresh._graph is not None
eager = AstTrainStep(cfg, sd, seed=3)
b = run(eager)
for i, ((la, oa), (lb, ob)) in enumerate(zip(a, b)):
    assert la == lb and torch.equal(oa, ob), (i, la, lb)
assert torch.equal(graph.p, eager.p)
print("GROWING-SHAPES-OK")
"""


def test_captured_step_when_the_shape_grows_in_a_fresh_process():
    """The captured trainer alone in a new process, stepping at a small shape first and larger ones later, with a
    second trainer growing the library's shared scratch under its live capture: every step must equal the eager
    trainer's bit for bit (a capture at a new shape follows one eager step there; a moved scratch buffer drops the
    recorded step; a capture that would have to grow the scratch is refused and the step runs eagerly)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GROWING_SHAPES_SCRIPT, root], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "GROWING-SHAPES-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_resume_keeps_the_optimisers_own_step_count():
    """Adam's bias correction counts the optimiser's updates (jt.optim.Adam's n_step), not the caller's iteration: a
    trainer started at current_iter = 5000 with fresh moments makes the first-update step, and a trainer resumed from
    optimizer_state_dict() continues bit-identically."""
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 6)
    lq, gt = _data(1, 128, 90)
    lq, gt = torch.from_numpy(lq), torch.from_numpy(gt)
    a = AstTrainStep(cfg, sd, drop_path_rate=0.0, use_grad_clip=False)
    b = AstTrainStep(cfg, sd, drop_path_rate=0.0, use_grad_clip=False)
    a.optimize_parameters(lq, gt, 1)
    b.optimize_parameters(lq, gt, 5000)
    assert a.opt_step == b.opt_step == 1 and torch.equal(a.p, b.p)
    a.optimize_parameters(lq, gt, 2)
    c = AstTrainStep(cfg, a.state_dict(), drop_path_rate=0.0, use_grad_clip=False)
    c.load_optimizer_state_dict(a.optimizer_state_dict())
    a.optimize_parameters(lq, gt, 3)
    c.optimize_parameters(lq, gt, 3)
    assert c.opt_step == 3 and torch.equal(a.p, c.p) and torch.equal(a.m, c.m) and torch.equal(a.v, c.v)


def test_gradient_mean_over_two_shards_equals_the_batch_gradient():
    """What the data-parallel step relies on: with per-image semantics the gradient of the mean loss over a batch is
    the mean of the gradients of its shards (here the two halves a 2-rank job would hold), so averaging the flat
    gradient buffers over the ranks reproduces the single-GPU step on the global batch."""
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 4)
    lq, gt = _data(2, 128, 21)
    tr = AstTrainStep(cfg, sd, coupled=False, drop_path_rate=0.0)
    loss, _ = tr.forward_backward(torch.from_numpy(lq), torch.from_numpy(gt))
    g_all = tr.g.clone()
    parts, losses = [], []
    for i in range(2):
        tr.g.zero_()
        li, _ = tr.forward_backward(torch.from_numpy(lq[i:i + 1]), torch.from_numpy(gt[i:i + 1]))
        parts.append(tr.g.clone())
        losses.append(li)
    mean = 0.5 * (parts[0] + parts[1])
    scale = float(g_all.abs().max())
    assert float((mean - g_all).abs().max()) < 1e-5 * scale
    assert abs(0.5 * sum(losses) - loss) < 1e-6 * abs(loss)


def test_a_few_steps_reduce_the_loss():
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 2)
    lq, gt = _data(1, 128, 9)
    tr = AstTrainStep(cfg, sd, lr=1e-3, use_grad_clip=False, drop_path_rate=0.0)
    losses = [tr.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt)) for _ in range(6)]
    print("[AST training step] losses on one fixed batch:", " ".join(f"{x:.5f}" for x in losses))
    assert losses[-1] < losses[0]


def test_model_wrapper_runs_the_step_from_the_option_tree(tmp_path):
    """ImageCleanModel with `is_train: true` and the train block of Enhancement/Options/AST_LOL_v2_synthetic.yml:
    feed_train_data + optimize_parameters (image_restoration_model.py:222-276) equals AstTrainStep driven directly,
    validation afterwards sees the updated weights, and what is not provided fails loudly."""
    from basicsr.models import create_model
    from jirf_hip import checkpoint
    from jirf_hip.train import AstTrainStep
    cfg = _cfg()
    sd = _state(cfg, 8)
    ck = tmp_path / "net_g.jth"
    checkpoint.save({"params": sd}, str(ck))
    opt = {"name": "t", "model_type": "ImageCleanModel", "is_train": True, "scale": 1, "manual_seed": 100,
           "network_g": dict(cfg, type="AST"),
           "path": {"pretrain_network_g": str(ck), "strict_load_g": True},
           "train": {"total_iter": 150000, "use_grad_clip": True, "mixing_augs": {"mixup": False},
                     "optim_g": {"type": "Adam", "lr": 2e-4, "betas": [0.9, 0.999]},
                     "pixel_opt": {"type": "L1Loss", "loss_weight": 1, "reduction": "mean"}}}
    model = create_model(opt)
    ref = AstTrainStep(cfg, sd, lr=2e-4, use_grad_clip=True, seed=100)     # the option's manual_seed
    for it in (1, 2):
        lq, gt = _data(2, 128, 30 + it)
        model.feed_train_data({"lq": torch.from_numpy(lq), "gt": torch.from_numpy(gt)})
        model.optimize_parameters(it)
        want = ref.optimize_parameters(torch.from_numpy(lq), torch.from_numpy(gt), it)
        assert abs(model.get_current_log()["l_pix"] - want) < 1e-9
    assert torch.equal(model.train_step.p, ref.p)                       # same kernels, same order: bit-identical
    lq, _ = _data(1, 128, 40)
    model.feed_data({"lq": torch.from_numpy(lq)})
    model.nonpad_test()
    _, pred = ref.forward_backward(torch.from_numpy(lq), torch.from_numpy(lq), backward=False)
    assert max_abs(model.output.cpu().numpy(), pred.cpu().numpy()) < 2e-5
    model.save_network(model.net_g, str(tmp_path / "net_g_2.jth"))
    back = checkpoint.load(str(tmp_path / "net_g_2.jth"))["params"]
    k = "decoderlayer_3.blocks.1.attn.w"
    assert np.array_equal(np.asarray(back[k]), ref.state_dict()[k]) and not np.array_equal(np.asarray(back[k]), sd[k])
    bad = dict(opt, network_g=dict(type="HINT"), path={})
    with pytest.raises(NotImplementedError):
        create_model(bad)
