"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the reference's validation arithmetic, used by
tests/ to check the GPU metric path.  Parity unpinned: the reference holds no fixture for these functions.

Follows basicsr/utils/img_util.py:136-166 (tensor2img), basicsr/metrics/metric_util.py:34-47 with
basicsr/utils/matlab_functions.py:268-282, 371-413 (to_y_channel / bgr2ycbcr) and
basicsr/metrics/psnr_ssim.py:52-69 (calculate_psnr)."""
import numpy as np


def tensor2img(chw: np.ndarray, rgb2bgr: bool = True) -> np.ndarray:
    """[C,H,W] float32 -> HxWxC (HxW for one channel) uint8."""
    t = np.clip(chw.astype(np.float32), 0.0, 1.0)
    img = t.transpose(1, 2, 0)
    if img.shape[2] == 1:
        img = img[:, :, 0]
    elif rgb2bgr:
        img = img[:, :, ::-1]
    return np.round(img * np.float32(255.0)).astype(np.uint8)


def to_y_channel(img: np.ndarray) -> np.ndarray:
    img = img.astype(np.float32) / 255.0
    if img.ndim == 3 and img.shape[2] == 3:
        y = np.dot(img, [24.966, 128.553, 65.481]) + 16.0      # float64: float32 image times a float64 list
        y /= 255.0
        img = y.astype(np.float32)[..., None]
    return img * 255.0


def calculate_psnr(img1: np.ndarray, img2: np.ndarray, crop_border: int = 0, test_y_channel: bool = False) -> float:
    """HWC (or HW) arrays."""
    assert img1.shape == img2.shape
    if img1.ndim == 2:
        img1, img2 = img1[..., None], img2[..., None]
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        img1 = img1[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img1, img2 = to_y_channel(img1), to_y_channel(img2)
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    max_value = 1.0 if img1.max() <= 1 else 255.0
    return 20.0 * np.log10(max_value / np.sqrt(mse))
