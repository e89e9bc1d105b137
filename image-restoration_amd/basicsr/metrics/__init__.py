"""Metric registry of the reference (basicsr/metrics/__init__.py): validation looks metrics up by the
`type` string of the YAML with getattr on this module (image_restoration_model.py:390-400)."""
from .psnr_ssim import calculate_psnr, calculate_ssim

__all__ = ["calculate_psnr", "calculate_ssim"]
