"""The reference's `ImageCleanModel` (basicsr/models/image_restoration_model.py:125-156, 230-233, 278-429): build
network_g from the option tree, load its checkpoint, run `pad_test` / `nonpad_test` over a data loader and
average the metrics the YAML names.  The network runs in libjirf_hip.so; inputs are moved to the GPU once and
the restored image stays there until a metric or an image file needs it.

Of the training half (:158-276) the optimisation step is provided for `network_g: {type: AST}` with the shipped
settings (pixel loss L1Loss = smooth-L1 mean, Adam, use_grad_clip): `is_train: true` builds a
jirf_hip.train.AstTrainStep and `optimize_parameters` runs forward, loss, backward, the gradient average over
the ranks, clip and update on the GPU.  Mixing augmentation, the FFT loss, EMA and the learning-rate schedulers
are not provided (set `lr` per iteration through `set_learning_rate`)."""
from __future__ import annotations

import logging
import os.path as osp
from collections import OrderedDict
from copy import deepcopy
from functools import partial

import numpy as np
import torch

from basicsr import metrics as metric_module
from basicsr.metrics.psnr_ssim import psnr_from_stats, psnr_stats
from basicsr.models.archs import define_network
from basicsr.utils.img_util import imwrite, tensor2img
from jirf_hip import checkpoint, harness

logger = logging.getLogger("basicsr")


class ImageCleanModel:
    def __init__(self, opt):
        self.opt = opt
        self.is_train = bool(opt.get("is_train", False))
        self.device = torch.device("cuda", int(opt.get("device", 0)))
        self.net_g = define_network(deepcopy(opt["network_g"])).cuda(self.device.index)
        path_opt = opt.get("path") or {}
        load_path = path_opt.get("pretrain_network_g", None)
        if load_path is not None:
            self.load_network(self.net_g, load_path, path_opt.get("strict_load_g", True),
                              param_key=path_opt.get("param_key", "params"))
        self.net_g.eval()
        self.metric_results = {}
        if self.is_train:
            self.init_training_settings()

    # ---- training (:158-276), AST only -------------------------------------------------------------------------
    def init_training_settings(self):
        from jirf_hip.train import AstTrainStep
        net_opt = dict(self.opt["network_g"])
        if net_opt.pop("type") != "AST":
            raise NotImplementedError("ImageCleanModel: the optimisation step is provided for network_g.type == 'AST' only")
        train_opt = self.opt["train"]
        if train_opt.get("ema_decay", 0) > 0 or train_opt.get("fft_loss_opt") or \
                (train_opt.get("mixing_augs") or {}).get("mixup", False):
            raise NotImplementedError("ImageCleanModel: EMA, the FFT loss and mixing augmentation are not provided")
        pix = dict(train_opt.get("pixel_opt") or {})
        if pix.get("type") != "L1Loss" or pix.get("reduction", "mean") != "mean":
            raise ValueError("pixel loss are None." if not pix else f"pixel loss {pix} is not provided (L1Loss, mean)")
        if not self.net_g.state_dict():
            # no checkpoint to start from: seeded random weights in the reference's key layout (the reference draws
            # its initial weights from Jittor's generator, whose stream cannot be reproduced here)
            from jirf_hip.synthetic import make_state
            self.net_g.load_state_dict(make_state("AST", net_opt, seed=int(self.opt.get("manual_seed", 0) or 0)), strict=False)
        optim = dict(train_opt["optim_g"])
        if optim.pop("type") != "Adam" or optim.get("weight_decay", 0):
            raise NotImplementedError("optimizer is not supperted yet (Adam without weight decay is)")
        self.train_step = AstTrainStep(net_opt, self.net_g.state_dict(), device=self.device.index,
                                       lr=float(optim.get("lr", 1e-3)), betas=tuple(optim.get("betas", (0.9, 0.999))),
                                       eps=float(optim.get("eps", 1e-8)),
                                       use_grad_clip=bool(train_opt.get("use_grad_clip", False)),
                                       loss_weight=float(pix.get("loss_weight", 1.0)),
                                       # basicsr/train.py:63-67 seeds every process with manual_seed + rank
                                       seed=int(self.opt.get("manual_seed", 0) or 0) + self._rank(),
                                       # a step is replayed as one captured hipGraph from its second run on
                                       use_graph=bool(train_opt.get("use_graph", True)))
        self.train_step.broadcast_parameters()
        self.log_dict = OrderedDict()

    @staticmethod
    def _rank() -> int:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def set_learning_rate(self, lr: float):
        self.train_step.lr = float(lr)

    def feed_train_data(self, data):                 # :222-228 (no mixing augmentation)
        self.feed_data(data)

    def optimize_parameters(self, current_iter):     # :235-276
        self.train_step.optimize_parameters(self.lq, self.gt, current_iter, sync_loss=False)
        self.output = self.train_step.output
        self.log_dict = OrderedDict(l_pix=self.train_step.log_loss)     # still on the device: read when it is logged
        self._net_is_stale = True

    def get_current_log(self):
        return OrderedDict((k, float(v.item()) if isinstance(v, torch.Tensor) else v) for k, v in self.log_dict.items())

    def sync_network(self):
        """Copies the trained weights into net_g (validation and save_network read them from there)."""
        if getattr(self, "_net_is_stale", False):
            self.net_g.load_state_dict(self.train_step.state_dict())
            self.net_g.cuda(self.device.index).eval()
            self._net_is_stale = False

    # base_model.py:228-258
    def load_network(self, net, load_path, strict=True, param_key="params"):
        return checkpoint.load_network(net, load_path, strict, param_key)

    def save_network(self, net, save_path, param_key="params"):
        if self.is_train and net is self.net_g:
            self.sync_network()
        return checkpoint.save_network(net, save_path, param_key)

    def _to_device(self, x) -> torch.Tensor:
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        if t.dim() == 3:
            t = t.unsqueeze(0)
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def feed_data(self, data):                       # :230-233
        self.lq = self._to_device(data["lq"])
        if "gt" in data:
            self.gt = self._to_device(data["gt"])

    def pad_test(self, window_size):                 # :278-291
        scale = self.opt.get("scale", 1)
        _, _, h, w = self.lq.shape
        mod_pad_h = window_size - h % window_size if h % window_size != 0 else 0
        mod_pad_w = window_size - w % window_size if w % window_size != 0 else 0
        img = harness.reflect_pad(self.lq, mod_pad_h, mod_pad_w)
        self.nonpad_test(img)
        _, _, h, w = self.output.shape
        self.output = self.output[:, :, 0:h - mod_pad_h * scale, 0:w - mod_pad_w * scale].contiguous()

    def nonpad_test(self, img=None):                 # :293-310
        if img is None:
            img = self.lq
        if self.is_train:
            self.sync_network()
        self.net_g.eval()
        pred = self.net_g(img)
        if isinstance(pred, (list, tuple)):
            pred = pred[-1]
        self.output = pred

    def get_current_visuals(self):                   # :423-429 (tensors stay on the GPU)
        out = OrderedDict(lq=self.lq, result=self.output)
        if hasattr(self, "gt"):
            out["gt"] = self.gt
        return out

    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False, rgb2bgr=True, use_image=True):
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image)

    def _metric(self, metric_type, result, gt, rgb2bgr, use_image, opt_):
        """One metric of one image pair ([1,C,H,W] on the GPU).  calculate_psnr is fused with the uint8
        conversion of tensor2img; any other name goes through the module lookup of the reference (:390-400)."""
        if metric_type == "calculate_psnr" and set(opt_) <= {"crop_border", "test_y_channel", "input_order"}:
            y = bool(opt_.get("test_y_channel", False))
            # as uint8 images the planes have been swapped to BGR when rgb2bgr is on (blue = plane 2 of the
            # RGB tensor); the float tensors are handed over as they are and plane 0 is taken for blue
            blue = 2 if (use_image and rgb2bgr) else 0
            return psnr_from_stats(*psnr_stats(result[0], gt[0], int(opt_.get("crop_border", 0)), bool(use_image), y, blue))
        fn = getattr(metric_module, metric_type)
        if use_image:
            return fn(tensor2img([result], rgb2bgr=rgb2bgr), tensor2img([gt], rgb2bgr=rgb2bgr), **opt_)
        return fn(result, gt, **opt_)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img, rgb2bgr, use_image):   # :322-411
        dataset = getattr(dataloader, "dataset", None)
        dataset_name = getattr(dataset, "opt", {}).get("name", "val") if dataset is not None else "val"
        val_opt = self.opt.get("val") or {}
        with_metrics = val_opt.get("metrics") is not None
        if with_metrics:
            self.metric_results = {metric: 0 for metric in val_opt["metrics"].keys()}
        window_size = val_opt.get("window_size", 0)
        test = partial(self.pad_test, window_size) if window_size else self.nonpad_test
        cnt = 0
        for val_data in dataloader:
            lq_path = val_data.get("lq_path", f"{cnt:06d}")
            lq_path = lq_path[0] if isinstance(lq_path, (list, tuple)) else lq_path
            img_name = osp.splitext(osp.basename(lq_path))[0]
            self.feed_data(val_data)
            test()
            visuals = self.get_current_visuals()
            if save_img:
                vis_dir = (self.opt.get("path") or {}).get("visualization", "visualization")
                imwrite(tensor2img([visuals["result"]], rgb2bgr=rgb2bgr), osp.join(vis_dir, dataset_name, f"{img_name}.png"))
                if "gt" in visuals:
                    imwrite(tensor2img([visuals["gt"]], rgb2bgr=rgb2bgr), osp.join(vis_dir, dataset_name, f"{img_name}_gt.png"))
            if with_metrics:
                for name, opt_ in deepcopy(val_opt["metrics"]).items():
                    metric_type = opt_.pop("type")
                    self.metric_results[name] += self._metric(metric_type, visuals["result"], visuals["gt"], rgb2bgr,
                                                              use_image, opt_)
            if hasattr(self, "gt"):
                del self.gt
            del self.lq
            del self.output
            cnt += 1
        current_metric = 0.0
        if with_metrics:
            for metric in self.metric_results.keys():
                self.metric_results[metric] /= max(cnt, 1)
                current_metric = self.metric_results[metric]
            log_str = f"Validation {dataset_name},\t" + "".join(f"\t # {m}: {v:.4f}" for m, v in self.metric_results.items())
            logger.info(log_str)
        return current_metric
