"""`tensor2img` / `imwrite` of the reference (basicsr/utils/img_util.py:98-170) for network outputs that
live on the GPU: clamp, scale and rounding happen in one kernel (jirf_op_crop_clamp_u8), only the 8-bit
image crosses to the host."""
from __future__ import annotations

import os

import numpy as np
import torch

from jirf_hip import harness


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    """[1,C,H,W] / [C,H,W] / [H,W] float tensor(s) on the GPU -> HxWxC (or HxW) uint8 array(s), channels BGR
    when rgb2bgr.  Only the form every caller in the reference uses is provided: uint8 output, range (0, 1)."""
    if not (isinstance(tensor, torch.Tensor) or (isinstance(tensor, list) and all(isinstance(t, torch.Tensor) for t in tensor))):
        raise TypeError(f"tensor or list of tensors expected, got {type(tensor)}")
    if out_type != np.uint8 or tuple(min_max) != (0, 1):
        raise NotImplementedError("tensor2img: only out_type=np.uint8 with min_max=(0, 1) is provided")
    single = isinstance(tensor, torch.Tensor)
    result = []
    for t in ([tensor] if single else tensor):
        if t.dim() == 4 and t.shape[0] == 1:
            t = t.squeeze(0)                       # img_util.py:136
        if t.dim() == 2:
            t = t.unsqueeze(0)
        if t.dim() != 3:
            raise TypeError(f"Only support 4D (one image), 3D or 2D tensor. But received with dimension: {t.dim()}")
        if not t.is_cuda:
            raise TypeError("tensor2img: the tensor must live on the GPU (there is no host path)")
        img = harness.to_uint8_image(t.detach().float().unsqueeze(0), t.shape[1], t.shape[2])[0].cpu().numpy()
        if img.shape[2] == 1:
            img = img[:, :, 0]
        elif rgb2bgr:
            img = np.ascontiguousarray(img[:, :, ::-1])
        result.append(img)
    return result[0] if len(result) == 1 else result


def imwrite(img, file_path, params=None, auto_mkdir=True):
    """img_util.py `imwrite`: BGR (or gray) uint8 array -> image file."""
    from PIL import Image
    if auto_mkdir:
        os.makedirs(os.path.abspath(os.path.dirname(file_path)), exist_ok=True)
    arr = img[:, :, ::-1] if img.ndim == 3 else img
    Image.fromarray(np.ascontiguousarray(arr)).save(file_path)
    return True
