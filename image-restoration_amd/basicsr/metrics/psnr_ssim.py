"""`calculate_psnr` (basicsr/metrics/psnr_ssim.py:11-69) and `calculate_ssim` (:247-330) of the reference with
their sums taken on the GPU (jirf_op_psnr_stats / jirf_op_ssim_stats of libjirf_hip.so); same arguments, same
peak-value rule, float64 results."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from jirf_hip import check, lib

_FP = C.POINTER(C.c_float)


def _planar(img, input_order: str, device) -> torch.Tensor:
    """Any accepted image -> float32 [C,H,W] on the GPU.  Tensors are CHW (a leading batch axis of 1 is
    dropped), as psnr_ssim.py:43-50 treats a Var; arrays follow `input_order` as reorder_image does
    (metric_util.py:27-31)."""
    if isinstance(img, torch.Tensor):
        t = img.detach()
        if t.dim() == 4:
            t = t.squeeze(0)
        if t.dim() == 2:
            t = t.unsqueeze(0)
    else:
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[..., None]
        elif input_order == "CHW":
            a = a.transpose(1, 2, 0)
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    if t.dim() != 3:
        raise ValueError(f"expected an image, got shape {tuple(img.shape)}")
    return t.to(device=device, dtype=torch.float32).contiguous()


def psnr_from_stats(sse: float, peak_probe: float, count: float) -> float:
    """psnr_ssim.py:65-69."""
    mse = sse / count
    if mse == 0:
        return float("inf")
    max_value = 1.0 if peak_probe <= 1 else 255.0
    return 20.0 * math.log10(max_value / math.sqrt(mse))


def psnr_stats(a: torch.Tensor, b: torch.Tensor, crop_border: int = 0, quantise: bool = False, y_channel: bool = False,
               blue_plane: int = 0):
    """(sum of squared differences, max of `a`, number of values) of two float32 [C,H,W] GPU images."""
    if a.shape != b.shape:
        raise AssertionError(f"Image shapes are differnet: {tuple(a.shape)}, {tuple(b.shape)}.")
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 3):
        raise TypeError("psnr_stats: float32 [C,H,W] tensors on the GPU expected")
    a, b = a.contiguous(), b.contiguous()
    c, h, w = (int(d) for d in a.shape)
    with torch.cuda.device(a.device):
        work = torch.empty(lib().jirf_op_psnr_work_doubles(), dtype=torch.float64, device=a.device)
        stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        check(lib().jirf_op_psnr_stats(C.cast(a.data_ptr(), _FP), C.cast(b.data_ptr(), _FP), c, h, w, int(crop_border),
                                       int(bool(quantise)), int(bool(y_channel)), int(blue_plane),
                                       C.c_void_p(work.data_ptr()), stream))
        sse, peak, count = work[:3].cpu().tolist()
    return sse, peak, count


def calculate_psnr(img1, img2, crop_border, input_order="HWC", test_y_channel=False, device=None):
    """Reference signature (psnr_ssim.py:11-17).  img1 / img2: HWC or CHW arrays in [0, 255] or [0, 1], or
    CHW tensors.  With test_y_channel the image is taken as BGR in [0, 255] like to_y_channel does
    (metric_util.py:34-47), whatever it really holds."""
    if tuple(img1.shape) != tuple(img2.shape):
        raise AssertionError(f"Image shapes are differnet: {img1.shape}, {img2.shape}.")
    if input_order not in ["HWC", "CHW"]:
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if device is None:
        device = next((t.device for t in (img1, img2) if isinstance(t, torch.Tensor) and t.is_cuda),
                      torch.device("cuda", torch.cuda.current_device()))
    a, b = _planar(img1, input_order, device), _planar(img2, input_order, device)
    if test_y_channel and a.shape[0] != 3:
        raise NotImplementedError("test_y_channel is provided for 3-channel images only")
    return psnr_from_stats(*psnr_stats(a, b, crop_border, False, bool(test_y_channel), 0))


def ssim_stats(a: torch.Tensor, b: torch.Tensor, crop_border: int = 0, quantise: bool = False, y_channel: bool = False,
               blue_plane: int = 0, max_value: float = 255.0):
    """(sum of the SSIM map, number of map values) of two float32 [C,H,W] GPU images."""
    if a.shape != b.shape:
        raise AssertionError(f"Image shapes are differnet: {tuple(a.shape)}, {tuple(b.shape)}.")
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 3):
        raise TypeError("ssim_stats: float32 [C,H,W] tensors on the GPU expected")
    a, b = a.contiguous(), b.contiguous()
    c, h, w = (int(d) for d in a.shape)
    with torch.cuda.device(a.device):
        work = torch.empty(lib().jirf_op_ssim_work_doubles(c, h, w), dtype=torch.float64, device=a.device)
        stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        check(lib().jirf_op_ssim_stats(C.cast(a.data_ptr(), _FP), C.cast(b.data_ptr(), _FP), c, h, w, int(crop_border),
                                       int(bool(quantise)), int(bool(y_channel)), int(blue_plane), float(max_value),
                                       C.c_void_p(work.data_ptr()), stream))
        total, count = work[:2].cpu().tolist()
    return total, count


def calculate_ssim(img1, img2, crop_border, input_order="HWC", test_y_channel=False, device=None):
    """Reference signature (psnr_ssim.py:247-253).  Colour images go through the 3-D Gaussian window of _ssim_3d
    (the window also spans the colour planes), test_y_channel through the 2-D window on the BT.601 luma of an image
    taken as BGR in [0, 255]."""
    if tuple(img1.shape) != tuple(img2.shape):
        raise AssertionError(f"Image shapes are differnet: {img1.shape}, {img2.shape}.")
    if input_order not in ["HWC", "CHW"]:
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if device is None:
        device = next((t.device for t in (img1, img2) if isinstance(t, torch.Tensor) and t.is_cuda),
                      torch.device("cuda", torch.cuda.current_device()))
    a, b = _planar(img1, input_order, device), _planar(img2, input_order, device)
    if test_y_channel:
        if a.shape[0] != 3:
            raise NotImplementedError("test_y_channel is provided for 3-channel images only")
        total, count = ssim_stats(a, b, crop_border, False, True, 0, 255.0)
    else:
        _, peak, _ = psnr_stats(a, b, crop_border)                       # img1.max() over the cropped window (:317)
        total, count = ssim_stats(a, b, crop_border, False, False, 0, 1.0 if peak <= 1 else 255.0)
    return total / count
