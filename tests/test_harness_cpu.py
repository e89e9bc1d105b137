"""CPU tests of the script-level plumbing oracle (index logic of the reference's test scripts)."""
import numpy as np
import pytest

from oracle import harness_oracle as ho


@pytest.mark.parametrize("h,w", [(321, 481), (320, 480), (7, 9), (720, 1280), (460, 620)])
def test_pad_amounts_and_reflect(h, w):
    padh, padw = ho.pad_amounts(h, w)
    assert (h + padh) % 8 == 0 and (w + padw) % 8 == 0 and 0 <= padh < 8 and 0 <= padw < 8
    if h < 16:
        return
    x = np.arange(h * w, dtype=np.float32).reshape(1, 1, h, w)
    y = ho.pad_to_factor(x)
    assert np.array_equal(y[:, :, :h, :w], x)
    for d in range(padh):     # row h + d mirrors row h - 2 - d (no edge repeat)
        assert np.array_equal(y[0, 0, h + d, :w], x[0, 0, h - 2 - d])
    for d in range(padw):
        assert np.array_equal(y[0, 0, :h, w + d], x[0, 0, :, w - 2 - d])


@pytest.mark.parametrize("L,crop,overlap", [(464, 256, 158), (624, 256, 158), (720, 256, 32), (1280, 256, 32),
                                            (256, 256, 158), (300, 128, 64), (129, 128, 64)])
def test_tile_starts_cover_and_end_flush(L, crop, overlap):
    st = ho.tile_starts(L, crop, overlap)
    assert st[-1] == L - crop and st[0] == 0 or L == crop
    assert all(0 <= s and s + crop <= L for s in st)
    covered = np.zeros(L, bool)
    for s in st:
        covered[s:s + crop] = True
    assert covered.all()
    assert st == sorted(st)


def test_split_merge_mean_blend():
    rng = np.random.default_rng(0)
    x = rng.random((2, 3, 72, 100), dtype=np.float32)
    split, starts = ho.splitimage(x, 32, 12)
    assert len(split) == len(starts) and split[0].shape == (2, 3, 32, 32)
    # tiles that agree where they overlap merge back to the image (up to the fp32 mean)
    y = ho.mergeimage(split, starts, 32, x.shape)
    assert np.abs(y - x).max() < 1e-6
    # different tiles: overlap regions are plain means in list order
    mod = [t + np.float32(i) for i, t in enumerate(split)]
    y = ho.mergeimage(mod, starts, 32, x.shape)
    cnt = np.zeros(x.shape, np.float32)
    acc = np.zeros(x.shape, np.float32)
    for i, (hs, ws) in enumerate(starts):
        acc[:, :, hs:hs + 32, ws:ws + 32] += mod[i]
        cnt[:, :, hs:hs + 32, ws:ws + 32] += 1
    assert np.array_equal(y, acc / cnt)


def test_uint8_rounding_is_half_to_even():
    v = np.array([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0, 0.999], np.float32).reshape(1, 1, 1, 6)
    got = ho.to_uint8(ho.crop_clamp(v, 1, 6))[0, 0, :, 0]
    want = np.rint(v.reshape(-1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(got, want)
    assert ho.to_uint8(ho.crop_clamp(np.full((1, 3, 2, 2), 1.7, np.float32), 2, 2)).max() == 255
    assert ho.to_uint8(ho.crop_clamp(np.full((1, 3, 2, 2), -0.2, np.float32), 2, 2)).max() == 0
