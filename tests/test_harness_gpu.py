"""Script-level plumbing kernels through the C ABI against the NumPy restatement: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import harness_oracle as ho

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,c,h,w", [(1, 3, 321, 481), (2, 3, 460, 620), (1, 3, 320, 480), (1, 1, 17, 23)])
def test_reflect_pad_bit_exact(b, c, h, w):
    from jirf_hip import harness as hh
    x = np.random.default_rng(1).random((b, c, h, w), dtype=np.float32)
    y = hh.pad_to_factor(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(y, ho.pad_to_factor(x))


@pytest.mark.parametrize("b,h,w,crop,overlap", [(1, 464, 624, 256, 158), (2, 72, 100, 32, 12), (1, 256, 256, 256, 158),
                                                 (1, 720, 1280, 256, 32)])
def test_split_and_merge_bit_exact(b, h, w, crop, overlap):
    from jirf_hip import harness as hh
    rng = np.random.default_rng(2)
    x = rng.random((b, 3, h, w), dtype=np.float32)
    split_ref, starts_ref = ho.splitimage(x, crop, overlap)
    split, starts = hh.splitimage(torch.from_numpy(x).cuda(), crop, overlap)
    assert starts == [tuple(int(v) for v in s) for s in starts_ref]
    for a, r in zip(split, split_ref):
        assert np.array_equal(a.cpu().numpy(), r)
    # merge tiles that differ in their overlaps (as restored tiles do)
    mod_ref = [t + np.float32(0.01 * i) for i, t in enumerate(split_ref)]
    mod = [torch.from_numpy(t).cuda() for t in mod_ref]
    y = hh.mergeimage(mod, starts, crop, x.shape).cpu().numpy()
    assert np.array_equal(y, ho.mergeimage(mod_ref, starts_ref, crop, x.shape))


def test_crop_clamp_and_uint8_bit_exact():
    from jirf_hip import harness as hh
    rng = np.random.default_rng(3)
    x = (rng.random((2, 3, 328, 488), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    x[0, 0, 0, :6] = np.array([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0, 0.999], np.float32)
    xd = torch.from_numpy(x).cuda()
    ref = ho.crop_clamp(x, 321, 481)
    assert np.array_equal(hh.crop_clamp(xd, 321, 481).cpu().numpy(), ref)
    assert np.array_equal(hh.to_uint8_image(xd, 321, 481).cpu().numpy(), ho.to_uint8(ref))


def test_restore_image_tiled_equals_manual_pipeline():
    """The composite (pad -> tiles -> network per tile batch -> mean blend -> unpad -> clamp) equals the
    same steps done one by one with the oracle's plumbing around the HIP network."""
    from basicsr.models.archs import define_network
    from jirf_hip import harness as hh
    from oracle.weights import make_astv2_state, synthetic_images
    from tests.make_golden import ASTV2_SMALL
    net = define_network(dict(ASTV2_SMALL, type="ASTv2"))
    net.load_state_dict(make_astv2_state(ASTV2_SMALL, seed=2, gain=0.6))
    net.cuda().eval()
    net.batch_mode = "per_image"
    x = synthetic_images(1, 83, 117, seed=5, sigma=25)
    got = hh.restore_image(net, torch.from_numpy(x).cuda(), crop_size=48, overlap_size=16, tile_batch=4).cpu().numpy()
    xp = ho.pad_to_factor(x)
    split, starts = ho.splitimage(xp, 48, 16)
    outs = [net(torch.from_numpy(np.ascontiguousarray(t)).cuda()).cpu().numpy() for t in split]
    want = ho.crop_clamp(ho.mergeimage(outs, starts, 48, xp.shape), 83, 117)
    assert np.array_equal(got, want)


def test_ast_enclosing_square_policy():
    """An image AST does not take (not square, side not a multiple of 128) is reflect-padded to the enclosing
    128-multiple square, restored and cropped; equals the oracle network on the same padded image."""
    from basicsr.models.archs import define_network
    from jirf_hip import harness as hh
    from jirf_hip.synthetic import make_state, synthetic_images
    from oracle import ast_oracle
    from tests.make_golden import AST_SHALLOW
    cfg = dict(AST_SHALLOW)
    sd = make_state("AST", cfg, seed=6)
    net = define_network(dict(cfg, type="AST"))
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synthetic_images(1, 100, 120, seed=8, sigma=15)
    assert hh.square_side(100, 120) == 128 and hh.square_side(321, 481) == 512 and hh.square_side(256, 256) == 256
    got = hh.restore_image(net, torch.from_numpy(x).cuda(), square_multiple=128).cpu().numpy()
    xp = np.pad(x, ((0, 0), (0, 0), (0, 28), (0, 8)), mode="reflect")
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    want = np.clip(ast_oracle.ast_forward(state, torch.from_numpy(xp), cfg).numpy()[:, :, :100, :120], 0, 1)
    assert got.shape == (1, 3, 100, 120)
    assert np.abs(got - want).max() < 1e-3
    with pytest.raises(ValueError, match="reflect padding"):
        hh.pad_to_square(torch.zeros(1, 3, 40, 200, device="cuda"))


def test_restore_images_over_a_folder_of_mixed_sizes():
    """harness.restore_images (the scripts' loop over a folder, shardable over ranks): with one rank every image
    comes back in list order and equals its own restore_image call - batches of equal-sized neighbours run with
    per-image semantics - for a stateless network and for HINT (every call cold)."""
    from basicsr.models.archs import define_network
    from jirf_hip import harness as hh
    from oracle.weights import make_astv2_state, make_hint_state, synthetic_images
    from tests.make_golden import ASTV2_SMALL, HINT_SMALL
    sizes = [(83, 117), (83, 117), (83, 117), (64, 96), (83, 117), (64, 96), (64, 96)]
    imgs = [torch.from_numpy(synthetic_images(1, h, w, seed=60 + i, sigma=25)[0]) for i, (h, w) in enumerate(sizes)]
    for arch, cfg, make in (("ASTv2", ASTV2_SMALL, make_astv2_state), ("HINT", HINT_SMALL, make_hint_state)):
        net = define_network(dict(cfg, type=arch))
        net.load_state_dict(make(cfg, seed=2, gain=0.6))
        net.cuda().eval()
        got = hh.restore_images(net, imgs, batch=2)
        kept = hh.restore_images(net, imgs, batch=3, balance=True, gather=False)
        assert sorted(kept) == list(range(len(imgs)))
        net.batch_mode = "per_image"
        for i, x in enumerate(imgs):
            if arch == "HINT":
                net.qv_cache = None
            want = hh.restore_image(net, x[None].cuda())[0]
            assert tuple(got[i].shape) == (3,) + sizes[i]
            assert float((got[i] - want).abs().max()) < 1e-5, (arch, i)
            assert float((kept[i] - want).abs().max()) < 1e-5, (arch, i)
