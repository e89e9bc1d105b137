"""One optimisation step of AST on the GPU: the device side of ImageCleanModel.optimize_parameters
(reference basicsr/models/image_restoration_model.py:235-276) for `network_g: {type: AST, token_mlp: frfn}`.

    zero_grad -> preds = net_g(lq) -> smooth-L1 mean (losses.py:12-14) -> backward -> [gradient average over
    the ranks] -> clip_grad_norm(0.01) (from the second iteration on) -> Adam step

Every operator - forward, data gradient, weight gradient, loss, clip, Adam - is a kernel of libjirf_hip.so
called through its C ABI (jirf_train_*); this module only owns device memory (torch tensors), records which
activations the backward needs, and walks the network in the reference's order (AST.execute
ast_arch.py:1220-1256, TransformerBlock.execute :768-842, FRFN.execute :476-513).  Tokens [B, L, C] of the
reference are planar [B, C, L] here, so Linear is a 1x1 convolution and the token LayerNorm a per-pixel one.

Parameters, gradients and the two Adam moments live in four flat fp32 buffers with one layout (every
parameter starts on a 256-byte boundary); the data-parallel step averages the gradient buffer over the ranks
with one all-reduce (`torch.distributed`, RCCL on GPUs) and nothing else is exchanged.

Stochastic depth (DropPath, ast_arch.py:52-70; rate 0.1 by the constructor default, per-block schedule
:981-983) is applied as in the reference's training mode: each branch of each block is multiplied per sample
by 0 or 1 / keep_prob.  The factors are drawn on the device from a counter-based generator keyed by `seed` (the
reference draws them from Jittor's generator, whose stream cannot be reproduced); `drop_scales` supplies them
explicitly instead (tests).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

import jirf_hip
from jirf_hip import check, lib

W_FWD, W_BWD_DATA, W_DECONV, W_DOWN_BWD, W_PACKED = 0, 1, 2, 4, 8
EP_STORE, EP_RESID, EP_GELU, EP_LEAKY = 0, 1, 6, 7

LAYERS = ["encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
          "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3"]
MULT = [1, 2, 4, 8, 16, 16, 8, 4, 2]
LEVEL = [0, 1, 2, 3, 4, 3, 2, 1, 0]


class V:
    """A planar activation view: device address of element (0, 0, 0) and the batch stride in floats."""
    __slots__ = ("ptr", "bs", "keep")

    def __init__(self, ptr: int, bs: int, keep=None):
        self.ptr, self.bs, self.keep = int(ptr), int(bs), keep

    def chan(self, c: int, P: int) -> "V":
        return V(self.ptr + 4 * c * P, self.bs, self.keep)


def _vp(a: Optional[int]):
    """A device address for a c_void_p parameter (ctypes converts ints and None itself)."""
    return a


def allreduce_flat_gradients(g: torch.Tensor, group=None) -> float:
    """The data-parallel exchange of the step: ONE all-reduce (sum) of the flat gradient buffer over the ranks of
    `group`, in place.  Returns 1 / ranks, the factor that turns the sum into the mean Jittor's optimizer takes under
    MPI; the training step hands it to the clip + Adam kernel (`grad_scale` of jirf_train_adam_step) instead of
    spending a pass over the buffer on it.  With an initialised process group the collective runs whatever the
    number of ranks - a one-rank launch under the launcher exercises the same RCCL call as eight - and without one
    nothing happens and the factor is 1.  Works on any device the group's backend serves (RCCL for GPU tensors,
    gloo for CPU tensors)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


def average_flat_gradients(g: torch.Tensor, group=None) -> None:
    """allreduce_flat_gradients followed by the division, for callers that want the mean in the buffer itself."""
    scale = allreduce_flat_gradients(g, group)
    if scale != 1.0:
        g.mul_(scale)


def broadcast_flat_parameters(p: torch.Tensor, group=None, src: int = 0) -> None:
    """The one-time parameter broadcast at start-up, so that every replica steps from the same weights."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(p, src=src, group=group)


def reduce_loss(loss, device, group=None):
    """reduce_loss_dict (basicsr/models/base_model.py:287-305): the logged loss is the mean over the ranks.  A float
    gives a float; a device tensor gives a device tensor (nothing waits for the value)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return loss
    lazy = isinstance(loss, torch.Tensor)
    t = loss.detach().clone().reshape(1) if lazy else torch.tensor([loss], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t / dist.get_world_size(group)
    return t[0] if lazy else float(t.item())


class AstTrainStep:
    """Holds the training state of one AST replica on one GPU and runs optimisation steps on it."""

    def __init__(self, cfg: dict, state: Dict[str, np.ndarray], device: int = 0, lr: float = 2e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, use_grad_clip: bool = True, max_norm: float = 0.01,
                 loss_weight: float = 1.0, coupled: bool = True, adam_form: str = "jittor", process_group=None,
                 drop_path_rate: Optional[float] = None, seed: int = 0, use_graph: bool = False):
        if cfg.get("token_mlp", "leff") != "frfn":
            raise NotImplementedError("AstTrainStep: only token_mlp='frfn' (the shipped AST configurations) is built")
        if not torch.cuda.is_available():
            raise RuntimeError("AstTrainStep needs a GPU: there is no CPU path")
        self.cfg = dict(cfg)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.L = lib()
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.use_grad_clip, self.max_norm, self.loss_weight = bool(use_grad_clip), float(max_norm), float(loss_weight)
        self.coupled = 1 if coupled else 0
        self.torch_form = 0 if adam_form == "jittor" else 1
        self.group = process_group
        self.iteration = 0        # the caller's current_iter: decides the clip (image_restoration_model.py:269)
        self.opt_step = 0         # the optimiser's own count of updates (jt.optim.Adam's n_step): bias correction
        self._grad_scale = 1.0    # 1 / ranks after the all-reduce of this step's gradient
        self.E = int(cfg.get("embed_dim", 32))
        self.depths = list(cfg.get("depths", [2] * 9))
        self.heads = list(cfg.get("num_heads", [1, 2, 4, 8, 16, 16, 8, 4, 2]))
        self.mlp_ratio = float(cfg.get("mlp_ratio", 4.0))
        self.qkv_bias = bool(cfg.get("qkv_bias", True))
        self.win = int(cfg.get("win_size", 8))
        self.img_size = int(cfg.get("img_size", 256))
        self.shift_flag = bool(cfg.get("shift_flag", True))
        self.dd_in = int(cfg.get("dd_in", 3))
        self.in_chans = int(cfg.get("in_chans", 3))
        # the parameter list (keys, shapes, order) is the library's, i.e. the reference's state dict
        spec_net = jirf_hip.Net("AST", cfg, device)
        self.specs = OrderedDict(spec_net.param_specs())
        spec_net.close()
        self.offset: Dict[str, int] = {}
        n = 0
        for k, shp in self.specs.items():
            self.offset[k] = n
            n += (int(np.prod(shp)) + 63) // 64 * 64
        self.n_flat = n
        self.p = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.load_state_dict(state)
        # stochastic depth: the option's rate (constructor default 0.1, ast_arch.py:955) unless given here
        self.drop_path_rate = float(cfg.get("drop_path_rate", 0.1) if drop_path_rate is None else drop_path_rate)
        self.drop_scales: Optional[Dict[str, np.ndarray]] = None      # explicit factors per branch (tests)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._draw_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._scale_B, self._scales, self._scales_drawn = -1, {}, {}
        self._stream = None
        # use_graph: optimize_parameters replays forward + backward as one captured hipGraph (from the second step on)
        self.use_graph = bool(use_graph)
        self._graph, self._graph_key, self._graph_in, self._graph_out = None, None, None, None
        self._graph_generation = -1
        self._eager_shapes = set()       # input shapes that have run a step outside a capture (scratch is sized)
        self._timing = False
        self.last_loss = None
        self._loss_ring = torch.zeros(64, dtype=torch.float64, device=self.device)
        self._loss_slot = -1
        self.loss_tensor = None
        # packed weights: (weight address, kind, Cin, Cout, ks) -> packed buffer; filled after the first full step
        self._pack_items: "OrderedDict[tuple, Optional[int]]" = OrderedDict()
        self._packed: Dict[tuple, int] = {}
        self._plan = None
        self._packed_buf = None

    def __del__(self):
        if getattr(self, "_plan", None) is not None:
            self.L.jirf_train_pack_plan_destroy(self._plan)
            self._plan = None

    def _build_pack_plan(self) -> None:
        """One launch per step writes every packed weight form (jirf_train_pack_plan_*), instead of one small
        kernel inside each convolution call.  The pairs are the ones the first step used."""
        keys = list(self._pack_items)
        n = len(keys)
        sizes = [int(self.L.jirf_train_packed_floats(k[1], k[2], k[3], k[4])) for k in keys]
        if n == 0 or min(sizes) <= 0:
            raise RuntimeError("pack plan: bad weight geometry")
        offs = np.concatenate([[0], np.cumsum([(s + 63) // 64 * 64 for s in sizes])])
        self._packed_buf = torch.empty(int(offs[-1]), dtype=torch.float32, device=self.device)
        base = self._packed_buf.data_ptr()
        wt = [base + 4 * int(o) for o in offs[:-1]]
        arr_p = (C.c_void_p * n)
        arr_i = (C.c_int * n)
        plan = C.c_void_p()
        check(self.L.jirf_train_pack_plan_create(
            n, arr_p(*[k[0] for k in keys]), arr_p(*[self._pack_items[k] for k in keys]), arr_p(*wt),
            arr_i(*[k[1] for k in keys]), arr_i(*[k[2] for k in keys]), arr_i(*[k[3] for k in keys]),
            arr_i(*[k[4] for k in keys]), C.byref(plan)))
        self._plan = plan
        self._packed = {k: p for k, p in zip(keys, wt)}

    # ---- parameters --------------------------------------------------------------------------------------------
    def load_state_dict(self, state: Dict[str, np.ndarray]) -> None:
        host = np.zeros(self.n_flat, dtype=np.float32)
        for k, shp in self.specs.items():
            if k not in state:
                raise KeyError(f"missing parameter {k}")
            a = np.asarray(state[k], dtype=np.float32)
            if tuple(a.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {a.shape}, expected {shp}")
            host[self.offset[k]: self.offset[k] + a.size] = a.reshape(-1)
        self.p.copy_(torch.from_numpy(host))

    def _unflatten(self, flat: torch.Tensor) -> "OrderedDict[str, np.ndarray]":
        host = flat.detach().cpu().numpy()
        return OrderedDict((k, host[self.offset[k]: self.offset[k] + int(np.prod(s))].reshape(s).copy())
                           for k, s in self.specs.items())

    def state_dict(self):
        return self._unflatten(self.p)

    def grad_dict(self):
        return self._unflatten(self.g)

    def num_parameters(self) -> int:
        return int(sum(int(np.prod(s)) for s in self.specs.values()))

    def P(self, key: str) -> int:
        return self.p.data_ptr() + 4 * self.offset[key]

    def G(self, key: str) -> int:
        return self.g.data_ptr() + 4 * self.offset[key]

    def Pb(self, key: str) -> Optional[int]:
        return self.P(key) if key in self.offset else None

    def Gb(self, key: str) -> Optional[int]:
        return self.G(key) if key in self.offset else None

    # ---- thin wrappers over the C ABI ------------------------------------------------------------------------------
    def _s(self):
        # looked up once per forward_backward (a captured step runs on the capture stream)
        return self._stream if self._stream is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def buf(self, B: int, Cn: int, P: int) -> V:
        t = torch.empty(B * Cn * P, dtype=torch.float32, device=self.device)
        return V(t.data_ptr(), Cn * P, t)

    def conv(self, x: V, w: int, kind: int, bias: Optional[int], r: Optional[V], y: V, B, Cin, Cout, Hin, Win, ks=1,
             stride=1, pad=0, epi=EP_STORE):
        # the packed form of (weight, kind) comes from the step's pack plan once the first step has listed the pairs
        key = (w, kind, Cin, Cout, ks)
        if self._plan is not None:
            wt = self._packed.get(key)
            if wt is not None:
                w, kind = wt, kind | W_PACKED
        else:
            self._pack_items.setdefault(key, bias if kind == W_DECONV else None)
        check(self.L.jirf_train_conv(_vp(x.ptr), x.bs, _vp(w), kind, _vp(bias), _vp(r.ptr if r else None),
                                     r.bs if r else 0, _vp(y.ptr), y.bs, B, Cin, Cout, Hin, Win, ks, stride, pad, epi,
                                     self._s()))

    def conv_dw(self, dy: V, x: V, dw: int, db: Optional[int], B, M, Cin, Hout, Wout, Hin, Win, ks=1, stride=1, pad=0):
        check(self.L.jirf_train_conv_bwd_weight(_vp(dy.ptr), dy.bs, _vp(x.ptr), x.bs, _vp(dw), _vp(db), B, M, Cin, Hout,
                                                Wout, Hin, Win, ks, stride, pad, 0, self._s()))

    def layernorm(self, x: V, key: str, y: V, B, Cn, P):
        check(self.L.jirf_train_layernorm(_vp(x.ptr), x.bs, _vp(self.P(key + ".weight")), _vp(self.P(key + ".bias")),
                                          _vp(y.ptr), y.bs, B, Cn, P, 1, self._s()))

    def layernorm_bwd(self, x: V, dy: V, key: str, add: Optional[V], dx: V, B, Cn, P):
        check(self.L.jirf_train_layernorm_bwd(_vp(x.ptr), x.bs, _vp(dy.ptr), dy.bs, _vp(self.P(key + ".weight")),
                                              _vp(add.ptr if add else None), add.bs if add else 0, _vp(dx.ptr), dx.bs,
                                              _vp(self.G(key + ".weight")), _vp(self.G(key + ".bias")), B, Cn, P, 1, 0,
                                              self._s()))

    def gelu(self, pre: V, x2: Optional[V], y: V, B, n_img, mode):
        check(self.L.jirf_train_gelu(_vp(pre.ptr), pre.bs, _vp(x2.ptr if x2 else None), x2.bs if x2 else 0, _vp(y.ptr),
                                     y.bs, B, n_img, mode, self._s()))

    def gelu_bwd(self, pre: V, x2: Optional[V], dy: V, dxa: V, dxb: Optional[V], B, n_img, mode):
        check(self.L.jirf_train_gelu_bwd(_vp(pre.ptr), pre.bs, _vp(x2.ptr if x2 else None), x2.bs if x2 else 0,
                                         _vp(dy.ptr), dy.bs, _vp(dxa.ptr), dxa.bs, _vp(dxb.ptr if dxb else None),
                                         dxb.bs if dxb else 0, B, n_img, mode, self._s()))

    def add(self, a: V, b: Optional[V], y: V, B, n_img):
        check(self.L.jirf_train_add(_vp(a.ptr), a.bs, _vp(b.ptr if b else None), b.bs if b else 0, _vp(y.ptr), y.bs, B,
                                    n_img, self._s()))

    def _draw_scales(self, B: int, launch: bool = True) -> None:
        """This step's DropPath factors (drop_path(), ast_arch.py:62-70: per sample 0 or 1 / keep_prob) for every
        branch with a non-zero rate, drawn on the device into a buffer that keeps its address (a captured step reads
        the same buffer at every replay), or taken from `drop_scales` when that is set."""
        if self.drop_scales is not None:
            rows, keys = [], []
            for key, f in self.drop_scales.items():
                f = torch.as_tensor(np.asarray(f, np.float32)).reshape(-1)
                if f.numel() != B:
                    raise ValueError(f"drop_scales[{key}]: {f.numel()} factors for a batch of {B}")
                rows.append(f)
                keys.append(key)
            d = torch.stack(rows).to(self.device) if rows else None
            self._scales = {k: d[i] for i, k in enumerate(keys)}
            return
        if self._scale_B != B:
            keys, rates = [], []
            for bl in self._blocks():
                for branch in (("attn", "mlp") if bl["att"] else ("mlp",)):
                    if bl["drop"] > 0.0:
                        keys.append(f"{bl['name']}.{branch}")
                        rates.append(bl["drop"])
            self._scale_B = B
            self._rate_dev = torch.tensor(rates, dtype=torch.float32, device=self.device)
            self._scale_dev = torch.zeros(max(len(keys), 1), B, dtype=torch.float32, device=self.device)
            self._scales_drawn = {k: self._scale_dev[i] for i, k in enumerate(keys)}
        if self._scales_drawn and launch:
            check(self.L.jirf_train_drop_path_scales(self._scale_dev.data_ptr(), self._rate_dev.data_ptr(),
                                                     len(self._scales_drawn), B, self.seed, self._draw_counter.data_ptr(),
                                                     self._s()))
        self._scales = self._scales_drawn

    def _branch_scale(self, bl, branch: str, B: int) -> Optional[torch.Tensor]:
        """DropPath factors of one branch for this step ([B] on the device), or None when the branch is kept."""
        return self._scales.get(f"{bl['name']}.{branch}")

    def scale_add(self, a: Optional[V], b: V, scale: torch.Tensor, y: V, B, n_img):
        check(self.L.jirf_train_scale_add(_vp(a.ptr if a else None), a.bs if a else 0, _vp(b.ptr), b.bs,
                                          _vp(scale.data_ptr()), _vp(y.ptr), y.bs, B, n_img, self._s()))

    # ---- one TransformerBlock (ast_arch.py:768-842) -------------------------------------------------------------
    def _blocks(self):
        out = []
        n_enc = sum(self.depths[:4])
        enc = [float(v) for v in torch.linspace(0, self.drop_path_rate, n_enc)] if n_enc > 1 else [0.0] * n_enc
        dec = enc[::-1]                                   # ast_arch.py:981-983
        e = k = 0
        for li in range(9):
            for j in range(self.depths[li]):
                Cn = self.E * MULT[li]
                win, shift = self.win, (self.win // 2 if (self.shift_flag and j % 2 == 1) else 0)
                if (self.img_size >> LEVEL[li]) <= self.win:
                    win, shift = self.img_size >> LEVEL[li], 0
                att = li >= 4
                if att and win != 8:
                    raise NotImplementedError("AstTrainStep: only 8 x 8 windows are built")
                if li < 4:
                    rate, e = enc[e], e + 1
                elif li == 4:
                    rate = self.drop_path_rate
                else:
                    if k >= len(dec):                     # the reference's constructor fails the same way
                        if self.drop_path_rate > 0.0:
                            raise IndexError("list index out of range (drop-path schedule: more decoder than encoder blocks)")
                        dec.append(0.0)
                    rate, k = dec[k], k + 1
                out.append(dict(name=f"{LAYERS[li]}.blocks.{j}", layer=li, C=Cn, hidden=int(Cn * self.mlp_ratio),
                                heads=self.heads[li], att=att, shift=shift, drop=rate))
        return out

    def _block_fwd(self, bl, x: V, out: V, B, H, W):
        n, Cn, h, P = bl["name"], bl["C"], bl["hidden"], H * W
        dc = Cn // 4
        sv = dict(x=x)
        cur = x
        if bl["att"]:
            a = n + ".attn"
            n1 = self.buf(B, Cn, P)
            self.layernorm(x, n + ".norm1", n1, B, Cn, P)
            qkv = self.buf(B, 3 * Cn, P)
            self.conv(n1, self.P(a + ".qkv.to_q.weight"), W_FWD, self.Pb(a + ".qkv.to_q.bias"), None, qkv, B, Cn, Cn, H, W)
            self.conv(n1, self.P(a + ".qkv.to_kv.weight"), W_FWD, self.Pb(a + ".qkv.to_kv.bias"), None,
                      qkv.chan(Cn, P), B, Cn, 2 * Cn, H, W)
            ao = self.buf(B, Cn, P)
            check(self.L.jirf_train_window_attention(_vp(qkv.ptr), qkv.bs, _vp(self.P(a + ".relative_position_bias_table")),
                                                     _vp(self.P(a + ".w")), _vp(ao.ptr), ao.bs, B, Cn, H, W, bl["heads"],
                                                     bl["shift"], self.coupled, self._s()))
            x1 = self.buf(B, Cn, P)
            sa = self._branch_scale(bl, "attn", B)
            if sa is None:
                self.conv(ao, self.P(a + ".proj.weight"), W_FWD, self.P(a + ".proj.bias"), x, x1, B, Cn, Cn, H, W, epi=EP_RESID)
            else:                                          # x + drop_path(branch)
                br = self.buf(B, Cn, P)
                self.conv(ao, self.P(a + ".proj.weight"), W_FWD, self.P(a + ".proj.bias"), None, br, B, Cn, Cn, H, W)
                self.scale_add(x, br, sa, x1, B, Cn * P)
            sv.update(n1=n1, qkv=qkv, ao=ao, x1=x1, s_attn=sa)
            cur = x1
        m = n + ".mlp"
        cat = self.buf(B, Cn, P)                       # norm2(x), then its first C/4 channels replaced by the 3x3's output
        self.layernorm(cur, n + ".norm2", cat, B, Cn, P)
        first = self.buf(B, dc, P)
        self.add(cat, None, first, B, dc * P)
        self.conv(first, self.P(m + ".partial_conv3.weight"), W_FWD, None, None, cat, B, dc, dc, H, W, 3, 1, 1)
        pre1 = self.buf(B, 2 * h, P)
        self.conv(cat, self.P(m + ".linear1.0.weight"), W_FWD, self.P(m + ".linear1.0.bias"), None, pre1, B, Cn, 2 * h, H, W)
        # GELU after linear1 (ast_arch.py:497): materialised only for the half the depthwise 3x3 reads; the gate applies
        # it to the other half on the fly (mode 3), which halves this pass and the one in the backward
        act1 = self.buf(B, h, P)
        self.gelu(pre1, None, act1, B, h * P, 0)
        predw = self.buf(B, h, P)
        check(self.L.jirf_train_dwconv3x3(_vp(act1.ptr), act1.bs, _vp(self.P(m + ".dwconv.0.weight")),
                                          _vp(self.P(m + ".dwconv.0.bias")), _vp(predw.ptr), predw.bs, B, h, H, W, 0,
                                          self._s()))
        gated = self.buf(B, h, P)
        self.gelu(predw, pre1.chan(h, P), gated, B, h * P, 3)
        sm = self._branch_scale(bl, "mlp", B)
        if sm is None:
            self.conv(gated, self.P(m + ".linear2.0.weight"), W_FWD, self.P(m + ".linear2.0.bias"), cur, out, B, h, Cn, H, W,
                      epi=EP_RESID)
        else:
            br = self.buf(B, Cn, P)
            self.conv(gated, self.P(m + ".linear2.0.weight"), W_FWD, self.P(m + ".linear2.0.bias"), None, br, B, h, Cn, H, W)
            self.scale_add(cur, br, sm, out, B, Cn * P)
        sv.update(cur=cur, cat=cat, first=first, pre1=pre1, act1=act1, predw=predw, gated=gated, s_mlp=sm)
        return sv

    def _block_bwd(self, bl, sv, dout: V, B, H, W) -> V:
        n, Cn, h, P = bl["name"], bl["C"], bl["hidden"], H * W
        dc = Cn // 4
        m = n + ".mlp"
        # linear2 (under stochastic depth the branch sees the output gradient times its per-sample factor)
        dbr = dout
        if sv.get("s_mlp") is not None:
            dbr = self.buf(B, Cn, P)
            self.scale_add(None, dout, sv["s_mlp"], dbr, B, Cn * P)
        self.conv_dw(dbr, sv["gated"], self.G(m + ".linear2.0.weight"), self.G(m + ".linear2.0.bias"), B, Cn, h, H, W, H, W)
        dgated = self.buf(B, h, P)
        self.conv(dbr, self.P(m + ".linear2.0.weight"), W_BWD_DATA, None, None, dgated, B, Cn, h, H, W)
        # gate and depthwise 3x3
        dact1 = self.buf(B, 2 * h, P)
        dpredw = self.buf(B, h, P)
        # (mode 3: the second half's gradient comes out already multiplied by the derivative of linear1's GELU)
        self.gelu_bwd(sv["predw"], sv["pre1"].chan(h, P), dgated, dpredw, dact1.chan(h, P), B, h * P, 3)
        check(self.L.jirf_train_dwconv3x3_bwd_weight(_vp(sv["act1"].ptr), sv["act1"].bs, _vp(dpredw.ptr), dpredw.bs,
                                                     _vp(self.G(m + ".dwconv.0.weight")), _vp(self.G(m + ".dwconv.0.bias")),
                                                     B, h, H, W, 0, self._s()))
        check(self.L.jirf_train_dwconv3x3(_vp(dpredw.ptr), dpredw.bs, _vp(self.P(m + ".dwconv.0.weight")), None,
                                          _vp(dact1.ptr), dact1.bs, B, h, H, W, 1, self._s()))
        # GELU after linear1 (in place; the first half only, see above), linear1
        self.gelu_bwd(sv["pre1"], None, dact1, dact1, None, B, h * P, 0)
        self.conv_dw(dact1, sv["cat"], self.G(m + ".linear1.0.weight"), self.G(m + ".linear1.0.bias"), B, 2 * h, Cn, H, W, H, W)
        dcat = self.buf(B, Cn, P)
        self.conv(dact1, self.P(m + ".linear1.0.weight"), W_BWD_DATA, None, None, dcat, B, 2 * h, Cn, H, W)
        # the dense 3x3 on the first C/4 channels
        self.conv_dw(dcat, sv["first"], self.G(m + ".partial_conv3.weight"), None, B, dc, dc, H, W, H, W, 3, 1, 1)
        dfirst = self.buf(B, dc, P)
        self.conv(dcat, self.P(m + ".partial_conv3.weight"), W_BWD_DATA, None, None, dfirst, B, dc, dc, H, W, 3, 1, 1)
        self.add(dfirst, None, dcat, B, dc * P)
        # norm2 and the shortcut around the MLP
        dcur = self.buf(B, Cn, P)
        self.layernorm_bwd(sv["cur"], dcat, n + ".norm2", dout, dcur, B, Cn, P)
        if not bl["att"]:
            return dcur
        a = n + ".attn"
        dbr = dcur
        if sv.get("s_attn") is not None:
            dbr = self.buf(B, Cn, P)
            self.scale_add(None, dcur, sv["s_attn"], dbr, B, Cn * P)
        self.conv_dw(dbr, sv["ao"], self.G(a + ".proj.weight"), self.G(a + ".proj.bias"), B, Cn, Cn, H, W, H, W)
        dao = self.buf(B, Cn, P)
        self.conv(dbr, self.P(a + ".proj.weight"), W_BWD_DATA, None, None, dao, B, Cn, Cn, H, W)
        dqkv = self.buf(B, 3 * Cn, P)
        qkv = sv["qkv"]
        check(self.L.jirf_train_window_attention_bwd(
            _vp(qkv.ptr), qkv.bs, _vp(self.P(a + ".relative_position_bias_table")), _vp(self.P(a + ".w")), _vp(dao.ptr),
            dao.bs, _vp(dqkv.ptr), dqkv.bs, _vp(self.G(a + ".relative_position_bias_table")), _vp(self.G(a + ".w")), B, Cn,
            H, W, bl["heads"], bl["shift"], self.coupled, 0, self._s()))
        self.conv_dw(dqkv, sv["n1"], self.G(a + ".qkv.to_q.weight"), self.Gb(a + ".qkv.to_q.bias"), B, Cn, Cn, H, W, H, W)
        self.conv_dw(dqkv.chan(Cn, P), sv["n1"], self.G(a + ".qkv.to_kv.weight"), self.Gb(a + ".qkv.to_kv.bias"), B, 2 * Cn,
                     Cn, H, W, H, W)
        dn1 = self.buf(B, Cn, P)
        self.conv(dqkv, self.P(a + ".qkv.to_q.weight"), W_BWD_DATA, None, None, dn1, B, Cn, Cn, H, W)
        self.conv(dqkv.chan(Cn, P), self.P(a + ".qkv.to_kv.weight"), W_BWD_DATA, None, dn1, dn1, B, 2 * Cn, Cn, H, W,
                  epi=EP_RESID)
        dx = self.buf(B, Cn, P)
        self.layernorm_bwd(sv["x"], dn1, n + ".norm1", dcur, dx, B, Cn, P)
        return dx

    # ---- the network (AST.execute ast_arch.py:1220-1256) -------------------------------------------------------------
    def forward_backward(self, lq: torch.Tensor, gt: torch.Tensor, backward: bool = True, sync_loss: bool = True):
        """Fills the gradient buffer with d loss / d parameters of this rank's batch and returns (loss, prediction).
        backward=False runs the forward alone, in evaluation mode (no stochastic depth).  sync_loss=False returns
        the loss as a 0-d float64 device tensor (valid for the next 64 steps) instead of waiting for it."""
        lq = lq.to(self.device, torch.float32).contiguous()
        gt = gt.to(self.device, torch.float32).contiguous()
        self._stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        try:
            loss_t, pred_t = self._forward_backward(lq, gt, backward, draw=True)
        finally:
            self._stream = None
        return (float(loss_t.item()) if sync_loss else loss_t), pred_t

    def _captured_forward_backward(self, lq: torch.Tensor, gt: torch.Tensor):
        """forward_backward(backward=True) as one hipGraph launch.  The step is a fixed sequence of several hundred
        kernels on fixed buffers, and queueing them one by one costs the host about as long as the GPU needs to run
        them; captured once per input shape (after the eager steps that size the scratch buffers and list the packed
        weights), it is replayed with the inputs copied into the captured step's buffers and the DropPath factors
        drawn into theirs.  Returns (loss tensor, prediction) of the captured step's buffers."""
        lq = lq.to(self.device, torch.float32)
        gt = gt.to(self.device, torch.float32)
        key = tuple(lq.shape)
        if self._graph is not None and self._graph_generation != self.L.jirf_train_scratch_generation():
            self._graph = None                            # some scratch buffer moved: the recorded addresses are stale
        if self._graph is None or self._graph_key != key:
            self._graph = None
            self._graph_in = (torch.empty_like(lq, memory_format=torch.contiguous_format),
                              torch.empty_like(gt, memory_format=torch.contiguous_format))
            self._draw_scales(key[0], launch=False)       # allocates the factor buffers outside the capture
            self._graph_loss = torch.zeros((), dtype=torch.float64, device=self.device)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                try:
                    self.g.zero_()
                    out = self._forward_backward(self._graph_in[0], self._graph_in[1], True, draw=False,
                                                 loss_out=self._graph_loss)
                finally:
                    self._stream = None
            self._graph, self._graph_key, self._graph_out = graph, key, out
            self._graph_generation = self.L.jirf_train_scratch_generation()
        self._graph_in[0].copy_(lq)
        self._graph_in[1].copy_(gt)
        self._draw_scales(key[0])
        self._graph.replay()
        loss_t, pred_t = self._graph_out
        self._loss_slot = (self._loss_slot + 1) % self._loss_ring.numel()
        self._loss_ring[self._loss_slot].copy_(loss_t)    # the captured step always writes the same slot
        self.loss_tensor = self._loss_ring[self._loss_slot]
        return self.loss_tensor, pred_t

    def _forward_backward(self, lq: torch.Tensor, gt: torch.Tensor, backward: bool, draw: bool,
                          loss_out: Optional[torch.Tensor] = None):
        B, Cin, H, W = lq.shape
        if Cin != self.dd_in or H != W or H % 128 != 0:
            raise ValueError(f"AST takes square inputs with side % 128 == 0 and {self.dd_in} channels, got {tuple(lq.shape)}")
        E = self.E
        Hl = [H >> l for l in range(5)]
        Pl = [(H >> l) * (W >> l) for l in range(5)]
        Cl = [E << l for l in range(5)]
        blocks = self._blocks()
        if not backward:
            self._scales = {}                             # forward only = evaluation mode: DropPath is the identity
        elif draw:
            self._draw_scales(B)
        if self._plan is not None:
            check(self.L.jirf_train_pack_plan_run(self._plan, self._s()))     # the weights may have changed since
        by_layer = [[b for b in blocks if b["layer"] == li] for li in range(9)]
        saved: Dict[str, dict] = {}
        xin = V(lq.data_ptr(), Cin * Pl[0], lq)

        def run_layer(li: int, x: V, Cn: int, final: Optional[V]) -> V:
            l = LEVEL[li]
            for j, bl in enumerate(by_layer[li]):
                last = j == len(by_layer[li]) - 1
                out = final if (last and final is not None) else self.buf(B, Cn, Pl[l])
                saved[bl["name"]] = self._block_fwd(bl, x, out, B, Hl[l], Hl[l])
                x = out
            return x

        cat = [self.buf(B, 2 * Cl[l], Pl[l]) for l in range(4)]
        y0 = self.buf(B, E, Pl[0])
        self.conv(xin, self.P("input_proj.proj.0.weight"), W_FWD, self.P("input_proj.proj.0.bias"), None, y0, B, Cin, E, H,
                  W, 3, 1, 1, EP_LEAKY)
        enc_in, enc_out = [y0], []
        for l in range(4):
            eo = run_layer(l, enc_in[l], Cl[l], cat[l].chan(Cl[l], Pl[l]))      # straight into the skip half of the concat
            enc_out.append(eo)
            nxt = self.buf(B, 2 * Cl[l], Pl[l + 1])
            k = f"dowsample_{l}.conv.0"
            self.conv(eo, self.P(k + ".weight"), W_FWD, self.P(k + ".bias"), None, nxt, B, Cl[l], 2 * Cl[l], Hl[l], Hl[l], 4,
                      2, 1)
            enc_in.append(nxt)
        below = run_layer(4, enc_in[4], Cl[4], None)
        belows = []
        for i in range(4):
            l = 3 - i
            cin_up = Cl[4] if i == 0 else 2 * Cl[l + 1]
            k = f"upsample_{i}.deconv.0"
            self.conv(below, self.P(k + ".weight"), W_DECONV, self.P(k + ".bias"), None, cat[l], B, cin_up, Cl[l], Hl[l + 1],
                      Hl[l + 1])
            belows.append((below, cin_up))
            below = run_layer(5 + i, cat[l], 2 * Cl[l], None)
        pred_t = torch.empty(B, self.in_chans, H, W, dtype=torch.float32, device=self.device)
        pred = V(pred_t.data_ptr(), self.in_chans * Pl[0], pred_t)
        self.conv(below, self.P("output_proj.proj.0.weight"), W_FWD, self.P("output_proj.proj.0.bias"),
                  xin if self.dd_in == 3 else None, pred, B, 2 * E, self.in_chans, H, W, 3, 1, 1,
                  EP_RESID if self.dd_in == 3 else EP_STORE)
        # ---- loss (losses.py:12-14: smooth-L1, mean) and its gradient ----------------------------------------------
        dY = self.buf(B, self.in_chans, Pl[0])
        # the loss stays on the device (a ring of slots, so that a value can be read many steps later); reading it on
        # the host here would stall the queue in the middle of the step
        if loss_out is None:
            self._loss_slot = (self._loss_slot + 1) % self._loss_ring.numel()
            loss_t = self._loss_ring[self._loss_slot]
        else:
            loss_t = loss_out
        check(self.L.jirf_train_smooth_l1(_vp(pred.ptr), _vp(gt.data_ptr()), _vp(dY.ptr if backward else None),
                                          pred_t.numel(), self.loss_weight, _vp(loss_t.data_ptr()), None, self._s()))
        self.loss_tensor = loss_t
        if not backward:
            return loss_t, pred_t
        # ---- backward ---------------------------------------------------------------------------------------------------
        def layer_bwd(li: int, d: V) -> V:
            l = LEVEL[li]
            for bl in reversed(by_layer[li]):
                d = self._block_bwd(bl, saved.pop(bl["name"]), d, B, Hl[l], Hl[l])
            return d

        self.conv_dw(dY, below, self.G("output_proj.proj.0.weight"), self.G("output_proj.proj.0.bias"), B, self.in_chans,
                     2 * E, H, W, H, W, 3, 1, 1)
        d = self.buf(B, 2 * E, Pl[0])
        self.conv(dY, self.P("output_proj.proj.0.weight"), W_BWD_DATA, None, None, d, B, self.in_chans, 2 * E, H, W, 3, 1, 1)
        dskip: List[Optional[V]] = [None] * 4
        for i in range(3, -1, -1):
            l = 3 - i
            dcat = layer_bwd(5 + i, d)                                   # [B, 2 C_l, P_l]: (upsampled | skip)
            dskip[l] = dcat.chan(Cl[l], Pl[l])
            below_in, cin_up = belows[i]
            k = f"upsample_{i}.deconv.0"
            check(self.L.jirf_train_channel_sum(_vp(dcat.ptr), dcat.bs, _vp(self.G(k + ".bias")), B, Cl[l], Pl[l], 0,
                                                self._s()))
            self.conv_dw(below_in, dcat, self.G(k + ".weight"), None, B, cin_up, Cl[l], Hl[l + 1], Hl[l + 1], Hl[l], Hl[l], 2,
                         2, 0)
            d = self.buf(B, cin_up, Pl[l + 1])
            self.conv(dcat, self.P(k + ".weight"), W_FWD, None, None, d, B, Cl[l], cin_up, Hl[l], Hl[l], 2, 2, 0)
        d = layer_bwd(4, d)
        for l in range(3, -1, -1):
            k = f"dowsample_{l}.conv.0"
            self.conv_dw(d, enc_out[l], self.G(k + ".weight"), self.G(k + ".bias"), B, 2 * Cl[l], Cl[l], Hl[l + 1], Hl[l + 1],
                         Hl[l], Hl[l], 4, 2, 1)
            denc = self.buf(B, Cl[l], Pl[l])
            self.conv(d, self.P(k + ".weight"), W_DOWN_BWD, None, None, denc, B, 2 * Cl[l], Cl[l], Hl[l + 1], Hl[l + 1])
            self.add(denc, dskip[l], denc, B, Cl[l] * Pl[l])
            d = layer_bwd(l, denc)
        dpre = self.buf(B, E, Pl[0])
        self.gelu_bwd(y0, None, d, dpre, None, B, E * Pl[0], 2)
        self.conv_dw(dpre, xin, self.G("input_proj.proj.0.weight"), self.G("input_proj.proj.0.bias"), B, E, Cin, H, W, H, W,
                     3, 1, 1)
        if self._plan is None:
            self._build_pack_plan()                       # the step has listed every (weight, kind) pair it packs
        return loss_t, pred_t

    # ---- the step (image_restoration_model.py:235-276) -------------------------------------------------------------
    def average_gradients(self) -> None:
        """The all-reduce (sum) of the gradient buffer; the division by the number of ranks rides in the update."""
        self._grad_scale = allreduce_flat_gradients(self.g, self.group)

    def broadcast_parameters(self, src: int = 0) -> None:
        broadcast_flat_parameters(self.p, self.group, src)

    def kernel_timing(self, on: bool) -> None:
        """Per-kernel device times (HIP events around every launch); while on, steps are queued kernel by kernel."""
        self._timing = bool(on)
        check(self.L.jirf_train_kernel_timing(int(on)))

    def kernel_times(self) -> List[dict]:
        cap = 1024
        names = (C.c_char_p * cap)()
        ms, fl, by = (C.c_float * cap)(), (C.c_double * cap)(), (C.c_double * cap)()
        ln = (C.c_int * cap)()
        n = C.c_int()
        check(self.L.jirf_train_kernel_times(names, ms, fl, by, ln, cap, C.byref(n)))
        return [dict(name=names[i].decode(), ms=float(ms[i]), flops=float(fl[i]), bytes=float(by[i]), launches=int(ln[i]))
                for i in range(min(n.value, cap))]

    def apply_update(self) -> None:
        """clip_grad_norm(0.01) when enabled and past the first iteration (image_restoration_model.py:269-270), then
        Adam (optimizer_g.step())."""
        clip = self.use_grad_clip and self.iteration > 1
        self.opt_step += 1
        if clip:
            check(self.L.jirf_train_sumsq(_vp(self.g.data_ptr()), self.n_flat, _vp(self._sumsq.data_ptr()), None, self._s()))
        check(self.L.jirf_train_adam_step(_vp(self.p.data_ptr()), _vp(self.g.data_ptr()), _vp(self.m.data_ptr()),
                                          _vp(self.v.data_ptr()), self.n_flat, _vp(self._sumsq.data_ptr()) if clip else None,
                                          self.max_norm if clip else 0.0, self._grad_scale, self.lr, self.betas[0],
                                          self.betas[1], self.eps, self.opt_step, self.torch_form, self._s()))
        self._grad_scale = 1.0

    def optimizer_state_dict(self) -> dict:
        """What a resume needs beside the weights: both Adam moments (per parameter, the state-dict layout) and the
        optimiser's step count (the reference pickles the optimizer object, base_model.py:260-285)."""
        return {"step": self.opt_step, "m": self._unflatten(self.m), "v": self._unflatten(self.v)}

    def load_optimizer_state_dict(self, state: dict) -> None:
        for name, flat in (("m", self.m), ("v", self.v)):
            host = np.zeros(self.n_flat, dtype=np.float32)
            for k, shp in self.specs.items():
                a = np.asarray(state[name][k], dtype=np.float32)
                if tuple(a.shape) != tuple(shp):
                    raise ValueError(f"{name}[{k}]: shape {a.shape}, expected {shp}")
                host[self.offset[k]: self.offset[k] + a.size] = a.reshape(-1)
            flat.copy_(torch.from_numpy(host))
        self.opt_step = int(state["step"])

    def optimize_parameters(self, lq: torch.Tensor, gt: torch.Tensor, current_iter: Optional[int] = None,
                            sync_loss: bool = True):
        """One iteration (image_restoration_model.py:235-276).  Returns this rank's loss: a float, read once the
        whole step has been queued, or with sync_loss=False the 0-d device tensor that will hold it (valid for the
        next 64 steps) - the reference's loss is a lazy Jittor variable too, evaluated when the logger prints it.
        `log_loss` is the mean over the ranks (reduce_loss_dict), of the same kind."""
        self.iteration = int(current_iter) if current_iter is not None else self.iteration + 1
        # a shape is captured only after a step at that shape has run outside a capture: that step grows the
        # library's scratch buffers to what the shape needs, which a capture cannot do
        shape = tuple(lq.shape)
        loss_t = None
        if self.use_graph and self._plan is not None and self.drop_scales is None and not self._timing and \
                shape in self._eager_shapes:
            try:
                loss_t, pred = self._captured_forward_backward(lq, gt)
            except jirf_hip.JirfError:
                # the capture was refused (scratch would have had to grow): this step runs kernel by kernel, which
                # sizes the buffers, and the next one captures again
                self._graph, self._stream = None, None
                torch.cuda.synchronize(self.device)
        if loss_t is None:
            self.g.zero_()
            loss_t, pred = self.forward_backward(lq, gt, sync_loss=False)
            self._eager_shapes.add(shape)
        self.output = pred
        self.average_gradients()
        self.apply_update()
        self.log_loss = reduce_loss(loss_t, self.device, self.group)
        if not sync_loss:
            return loss_t
        if isinstance(self.log_loss, torch.Tensor):
            self.log_loss = float(self.log_loss.item())
        self.last_loss = float(loss_t.item())
        return self.last_loss
