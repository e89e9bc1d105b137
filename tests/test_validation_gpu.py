"""The validation path of the reference's model class on the GPU: calculate_psnr, tensor2img, pad_test and
the metric loop (basicsr/models/image_restoration_model.py:278-411), against the numpy restatement in
oracle/metrics_oracle.py and the torch-CPU oracle of the network."""
import numpy as np
import pytest
import torch

from oracle import astv2_oracle, metrics_oracle as mo
from tests.make_golden import ASTV2_SMALL

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(5)


def _pair(c, h, w, lo=-0.1, hi=1.1):
    a = RNG.uniform(lo, hi, (c, h, w)).astype(np.float32)
    b = (a + RNG.normal(0, 0.05, a.shape)).astype(np.float32)
    return a, b


@pytest.mark.parametrize("c,h,w,border", [(3, 37, 52, 0), (3, 64, 64, 4), (1, 19, 23, 2)])
def test_psnr_float_tensors(c, h, w, border):
    from basicsr.metrics import calculate_psnr
    a, b = _pair(c, h, w)
    want = mo.calculate_psnr(a.transpose(1, 2, 0), b.transpose(1, 2, 0), border)
    got = calculate_psnr(torch.from_numpy(a).cuda()[None], torch.from_numpy(b).cuda()[None], border)
    assert abs(got - want) < 1e-9
    # peak-value rule (psnr_ssim.py:68): values within [0, 1] are scored against 1.0
    a2, b2 = np.clip(a, 0, 1), np.clip(b, 0, 1)
    want = mo.calculate_psnr(a2.transpose(1, 2, 0), b2.transpose(1, 2, 0), border)
    got = calculate_psnr(a2, b2, border, input_order="CHW")
    assert abs(got - want) < 1e-9 and want < 60
    assert calculate_psnr(a2, a2, 0, input_order="CHW") == float("inf")


@pytest.mark.parametrize("y", [False, True])
@pytest.mark.parametrize("rgb2bgr", [True, False])
def test_psnr_uint8_images_and_y_channel(y, rgb2bgr):
    from basicsr.metrics import calculate_psnr
    from basicsr.metrics.psnr_ssim import psnr_from_stats, psnr_stats
    from basicsr.utils import tensor2img
    a, b = _pair(3, 45, 61)
    ia, ib = mo.tensor2img(a, rgb2bgr), mo.tensor2img(b, rgb2bgr)
    want = mo.calculate_psnr(ia, ib, 3, test_y_channel=y)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ga = tensor2img(ta[None], rgb2bgr=rgb2bgr)
    assert ga.dtype == np.uint8 and np.array_equal(ga, ia)                 # bit-exact 8-bit image
    got = calculate_psnr(ga, tensor2img([tb], rgb2bgr=rgb2bgr), 3, test_y_channel=y)
    assert abs(got - want) < 1e-6
    fused = psnr_from_stats(*psnr_stats(ta, tb, 3, True, y, 2 if rgb2bgr else 0))
    assert abs(fused - want) < 1e-6


def test_psnr_argument_errors():
    from basicsr.metrics import calculate_psnr, calculate_ssim
    a = np.zeros((8, 8, 3), np.float32)
    with pytest.raises(AssertionError):
        calculate_psnr(a, np.zeros((8, 9, 3), np.float32), 0)
    with pytest.raises(ValueError):
        calculate_psnr(a, a, 0, input_order="WHC")
    with pytest.raises(Exception, match="crop_border"):
        calculate_psnr(a, a, 4)
    with pytest.raises(AssertionError):
        calculate_ssim(a, np.zeros((8, 9, 3), np.float32), 0)
    with pytest.raises(ValueError):
        calculate_ssim(a, a, 0, input_order="WHC")


@pytest.mark.parametrize("shape,border,y", [((45, 61, 3), 0, False), ((45, 61, 3), 3, False), ((64, 96, 3), 0, True),
                                            ((37, 33, 3), 4, True), ((70, 40, 1), 0, False)])
def test_ssim_matches_oracle(shape, border, y):
    """calculate_ssim (psnr_ssim.py:247-330): the 3-D window over H x W x C for colour images, the 2-D window on
    the luma with test_y_channel; uint8-range arrays, [0, 1] float arrays (max_value rule :317) and GPU tensors."""
    from basicsr.metrics import calculate_ssim
    rng = np.random.default_rng(sum(shape) + border)
    a = rng.integers(0, 256, shape).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-25, 26, shape), 0, 255).astype(np.uint8)
    if shape[2] == 1:
        a, b = a[..., 0], b[..., 0]
    want = mo.calculate_ssim(a, b, border, test_y_channel=y)
    got = calculate_ssim(a, b, border, test_y_channel=y)
    # the luma form is float64 end to end in the reference (_ssim_cly :222-223) and here
    assert abs(got - want) < (1e-10 if y else 1e-5), (got, want)
    assert abs(calculate_ssim(a, a, border, test_y_channel=y) - 1.0) < 1e-6
    if not y and a.ndim == 3:
        fa, fb = a.astype(np.float32) / 255.0, b.astype(np.float32) / 255.0          # [0, 1] images: C1, C2 from max 1
        want01 = mo.calculate_ssim(fa, fb, border)
        assert abs(calculate_ssim(fa, fb, border) - want01) < 1e-5
        ta = torch.from_numpy(np.ascontiguousarray(fa.transpose(2, 0, 1))).cuda()    # CHW tensor, as a Var is treated
        tb = torch.from_numpy(np.ascontiguousarray(fb.transpose(2, 0, 1))).cuda()
        assert abs(calculate_ssim(ta[None], tb[None], border) - want01) < 1e-5
        chw = calculate_ssim(a.transpose(2, 0, 1), b.transpose(2, 0, 1), border, input_order="CHW")
        assert abs(chw - want) < 1e-5


def test_ssim_luma_form_on_flat_regions_is_float64_exact():
    """Flat and nearly flat regions are where sigma = E[x^2] - mu^2 cancels: bright plateaus with +-1 grey level of
    noise.  In float32 the variances carry errors of about 4e-3 against C2 = 58.5; the float64 path agrees with the
    oracle's float64 evaluation to 1e-10."""
    from basicsr.metrics import calculate_ssim
    rng = np.random.default_rng(5)
    a = np.full((72, 88, 3), 250, np.uint8)
    a[:, 44:] = 17
    a[20:40] = rng.integers(0, 256, (20, 88, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-1, 2, a.shape), 0, 255).astype(np.uint8)
    want = mo.calculate_ssim(a, b, 0, test_y_channel=True)
    got = calculate_ssim(a, b, 0, test_y_channel=True)
    assert abs(got - want) < 1e-10, (got, want)


def test_validation_loop_matches_oracle(tmp_path):
    """create_model -> load .jth -> pad_test(window 8) -> PSNR, on images whose sides are not multiples
    of 8, against the oracle network + numpy metrics; then the same through `python -m basicsr.test`."""
    from PIL import Image
    from basicsr.models import create_model
    from basicsr import test as test_entry
    from jirf_hip import checkpoint, synthetic
    cfg = dict(ASTV2_SMALL)
    sd = synthetic.make_state("ASTv2", cfg, seed=4)
    ck = tmp_path / "net_g_latest.jth"
    checkpoint.save({"params": sd}, str(ck))
    (tmp_path / "gt").mkdir()
    (tmp_path / "lq").mkdir()
    pairs = []
    for i, (h, w) in enumerate([(37, 52), (40, 48)]):
        gt = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
        lq = np.clip(gt.astype(np.int32) + RNG.integers(-30, 31, gt.shape), 0, 255).astype(np.uint8)
        Image.fromarray(gt).save(tmp_path / "gt" / f"{i}.png")
        Image.fromarray(lq).save(tmp_path / "lq" / f"{i}.png")
        pairs.append((lq, gt))
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    want = {"img": [], "float": []}
    for lq, gt in pairs:
        x = torch.from_numpy(lq.astype(np.float32).transpose(2, 0, 1) / 255.0)[None]
        g = gt.astype(np.float32).transpose(2, 0, 1) / 255.0
        h, w = x.shape[2:]
        ph, pw = (8 - h % 8) % 8, (8 - w % 8) % 8
        xp = torch.nn.functional.pad(x, (0, pw, 0, ph), mode="reflect")
        out = astv2_oracle.astv2_forward(state, xp, cfg)[0, :, :h, :w].numpy()
        want["img"].append(mo.calculate_psnr(mo.tensor2img(out), mo.tensor2img(g), 0))
        want["float"].append(mo.calculate_psnr(out.transpose(1, 2, 0), g.transpose(1, 2, 0), 0))
    opt = {
        "name": "t", "model_type": "ImageCleanModel", "scale": 1,
        "datasets": {"test_1": {"name": "pairs", "type": "Dataset_PairedImage", "dataroot_gt": str(tmp_path / "gt"),
                                "dataroot_lq": str(tmp_path / "lq"), "io_backend": {"type": "disk"}}},
        "network_g": dict(cfg, type="ASTv2"),
        "path": {"pretrain_network_g": str(ck), "strict_load_g": True, "param_key": "params_ema",
                 "visualization": str(tmp_path / "vis")},
        "val": {"window_size": 8, "save_img": True, "rgb2bgr": True, "use_image": True,
                "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 0, "test_y_channel": False}}},
    }
    import yaml
    yml = tmp_path / "opt.yml"
    yml.write_text(yaml.safe_dump(opt))
    res = test_entry.main(["-opt", str(yml)])
    assert abs(res["pairs"]["psnr"] - np.mean(want["img"])) < 0.01          # |dPSNR| <= 0.01 dB
    saved = np.asarray(Image.open(tmp_path / "vis" / "pairs" / "0.png"))
    assert saved.shape == (37, 52, 3)
    # float-tensor metrics (use_image: false, what most option files set)
    opt["is_train"] = False
    model = create_model(opt)
    from basicsr.data import create_dataloader, create_dataset
    loader = create_dataloader(create_dataset(opt["datasets"]["test_1"]))
    got = model.validation(loader, "t", None, save_img=False, rgb2bgr=True, use_image=False)
    assert abs(got - np.mean(want["float"])) < 0.01
