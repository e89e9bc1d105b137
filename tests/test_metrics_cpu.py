"""Known answers for the numpy restatement of the reference's validation arithmetic
(oracle/metrics_oracle.py), plus the host glue of the test entry point; no GPU."""
import numpy as np
import pytest

from oracle import metrics_oracle as mo


def test_psnr_known_answers():
    a = np.full((16, 16, 3), 100, np.uint8)
    b = np.full((16, 16, 3), 110, np.uint8)
    assert mo.calculate_psnr(a, b) == pytest.approx(20 * np.log10(255 / 10), abs=1e-12)     # mse = 100, peak 255
    assert mo.calculate_psnr(a, a) == float("inf")
    fa, fb = np.full((8, 8, 3), 0.5, np.float32), np.full((8, 8, 3), 0.25, np.float32)
    assert mo.calculate_psnr(fa, fb) == pytest.approx(20 * np.log10(1 / 0.25), abs=1e-12)   # max <= 1: peak 1.0
    fa2 = fa.copy()
    fa2[0, 0, 0] = 1.5                                                                      # one value above 1: peak 255
    assert mo.calculate_psnr(fa2, fb) > 40
    # crop_border drops the rim
    c = a.copy()
    c[0, :, :] = 0
    assert mo.calculate_psnr(a, c, crop_border=1) == float("inf")


def test_y_channel_is_bt601_studio_range():
    white = np.full((4, 4, 3), 255, np.uint8)
    black = np.zeros((4, 4, 3), np.uint8)
    assert np.allclose(mo.to_y_channel(white.astype(np.float64)), 235.0, atol=1e-4)         # 16 + 219
    assert np.allclose(mo.to_y_channel(black.astype(np.float64)), 16.0, atol=1e-5)
    blue = np.zeros((1, 1, 3), np.float64)
    blue[..., 0] = 255                                                                      # BGR: plane 0 is blue
    assert np.allclose(mo.to_y_channel(blue), 16 + 24.966, atol=1e-4)


def test_tensor2img_rounding_and_order():
    t = np.array([[[0.5 / 255, 1.5 / 255, 2.5 / 255, -1.0, 2.0]]] * 3, np.float32)          # [3,1,5]
    t[2] += 1.0 / 255
    img = mo.tensor2img(t, rgb2bgr=True)
    assert img.dtype == np.uint8 and img.shape == (1, 5, 3)
    assert img[0, :, 2].tolist() == [0, 2, 2, 0, 255]          # red plane lands last; halves go to the even value
    assert img[0, 3].tolist() == [0, 0, 0] and img[0, 4].tolist() == [255, 255, 255]
    assert mo.tensor2img(t[:1]).shape == (1, 5)


def test_paired_folder_and_option_parsing(tmp_path):
    from PIL import Image
    import yaml
    from basicsr.data import create_dataloader, create_dataset
    from basicsr.test import parse_options
    (tmp_path / "gt").mkdir()
    (tmp_path / "lq").mkdir()
    rgb = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    for d in ("gt", "lq"):
        Image.fromarray(rgb).save(tmp_path / d / "a.png")
    (tmp_path / "gt" / "notes.txt").write_text("ignored")
    opt = {"name": "x", "scale": 1, "datasets": {"test_1": {"name": "p", "type": "Dataset_PairedImage",
           "dataroot_gt": str(tmp_path / "gt"), "dataroot_lq": str(tmp_path / "lq"), "io_backend": {"type": "disk"}}}}
    yml = tmp_path / "o.yml"
    yml.write_text(yaml.safe_dump(opt))
    parsed = parse_options(["-opt", str(yml)])
    assert parsed["is_train"] is False and parsed["datasets"]["test_1"]["phase"] == "test"
    loader = create_dataloader(create_dataset(parsed["datasets"]["test_1"]))
    items = list(loader)
    assert len(items) == 1 and items[0]["lq"].shape == (1, 3, 5, 7) and items[0]["lq"].dtype == np.float32
    assert np.array_equal(np.round(items[0]["gt"][0].transpose(1, 2, 0) * 255).astype(np.uint8), rgb)   # RGB order kept
    with pytest.raises(ValueError, match="is not found"):
        create_dataset({"type": "Dataset_GaussianDenoising"})
    from basicsr.models import create_model
    with pytest.raises(ValueError, match="is not found"):
        create_model({"model_type": "NoSuchModel"})


def test_ssim_oracle_known_answers():
    """Constant images have a closed form, (2 a b + C1) / (a^2 + b^2 + C1) (the covariance term is 1); identical
    images give 1; the colour form's window spans the colour planes (a change in one plane moves the others' maps)."""
    from oracle import metrics_oracle as mo
    a = np.full((24, 30, 3), 100, np.uint8)
    b = np.full((24, 30, 3), 140, np.uint8)
    c1 = (0.01 * 255) ** 2
    want = (2 * 100 * 140 + c1) / (100 ** 2 + 140 ** 2 + c1)
    assert abs(mo.calculate_ssim(a, b, 0) - want) < 1e-6
    assert abs(mo.calculate_ssim(a, b, 2, test_y_channel=False, dtype=np.float64) - want) < 1e-9
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (33, 41, 3)).astype(np.uint8)
    assert abs(mo.calculate_ssim(x, x, 0) - 1.0) < 1e-6 and abs(mo.calculate_ssim(x, x, 0, test_y_channel=True) - 1.0) < 1e-9
    y = x.copy()
    y[..., 0] = 255 - y[..., 0]
    full = mo.calculate_ssim(x, y, 0)
    per_plane = np.mean([mo.calculate_ssim(x[..., i], y[..., i], 0) for i in range(3)])
    assert abs(full - per_plane) > 1e-3
    g = mo._gaussian_11()
    assert abs(g.sum() - 1) < 1e-12 and abs(g[5] / g[4] - np.exp(1 / 4.5)) < 1e-12
