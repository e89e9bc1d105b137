"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the reference's validation arithmetic, used by
tests/ to check the GPU metric path.  Parity unpinned: the reference holds no fixture for these functions.

Follows basicsr/utils/img_util.py:136-166 (tensor2img), basicsr/metrics/metric_util.py:34-47 with
basicsr/utils/matlab_functions.py:268-282, 371-413 (to_y_channel / bgr2ycbcr) and
basicsr/metrics/psnr_ssim.py:52-69 (calculate_psnr), :145-244 and :296-330 (calculate_ssim)."""
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


def _gaussian_11() -> np.ndarray:
    """cv2.getGaussianKernel(11, 1.5) (psnr_ssim.py:146,221): exp(-(i - 5)^2 / (2 sigma^2)) normalised to sum 1."""
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2.0 * 1.5 ** 2))
    return g / g.sum()


def _window(vol: np.ndarray, axes) -> np.ndarray:
    """The separable Gaussian window along `axes`, replicate padding (Conv3d padding_mode="replicate" :157,
    cv2.BORDER_REPLICATE :225)."""
    from scipy.ndimage import correlate1d
    out = vol
    for ax in axes:
        out = correlate1d(out, _gaussian_11(), axis=ax, mode="nearest")
    return out


def _ssim_map(img1: np.ndarray, img2: np.ndarray, axes, max_value: float, dtype) -> float:
    c1, c2 = (0.01 * max_value) ** 2, (0.03 * max_value) ** 2
    a, b = img1.astype(dtype), img2.astype(dtype)
    mu1, mu2 = _window(a, axes), _window(b, axes)
    s1 = _window(a * a, axes) - mu1 * mu1
    s2 = _window(b * b, axes) - mu2 * mu2
    s12 = _window(a * b, axes) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(m.astype(np.float64).mean())


def calculate_ssim(img1: np.ndarray, img2: np.ndarray, crop_border: int = 0, test_y_channel: bool = False,
                   dtype=np.float32) -> float:
    """HWC arrays.  Colour form: the 11 x 11 x 11 window over the H x W x C volume (_ssim_3d :161-199, float32
    convolution); test_y_channel: the 2-D window on the luma (_ssim_cly :202-244, float64)."""
    assert img1.shape == img2.shape
    if img1.ndim == 2:
        img1, img2 = img1[..., None], img2[..., None]
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        img1 = img1[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        y1, y2 = to_y_channel(img1)[..., 0], to_y_channel(img2)[..., 0]
        return _ssim_map(y1, y2, (0, 1), 255.0, np.float64)
    max_value = 1.0 if img1.max() <= 1 else 255.0
    return _ssim_map(img1, img2, (0, 1, 2), max_value, dtype)
