"""CPU restatement (NumPy) of the image plumbing around the network call in the reference's test
scripts.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is imported by the product path.

Follows Deraining/test_rain100L.py (the same code is repeated in every <Task>/test_*.py):
  pad_to_factor      :147-152   reflect padding of the bottom / right border up to a multiple of 8
  tile_starts        :33-40     the start lists of splitimage
  splitimage         :31-48
  mergeimage         :65-75     with get_scoremap(is_mean=True) (:51-62): a map of ones
  crop_clamp, to_uint8  :171-190  unpad, clamp to [0, 1], NHWC, skimage.img_as_ubyte
Parity unpinned: the reference ships no fixtures for these helpers; they are index / byte
operations restated literally.
"""
from __future__ import annotations

import numpy as np


def pad_amounts(h: int, w: int, factor: int = 8):
    """:148-151 — H = ((h + factor) // factor) * factor; no padding when already a multiple."""
    H, W = ((h + factor) // factor) * factor, ((w + factor) // factor) * factor
    padh = H - h if h % factor != 0 else 0
    padw = W - w if w % factor != 0 else 0
    return padh, padw


def pad_to_factor(x: np.ndarray, factor: int = 8) -> np.ndarray:
    """:152 — nn.pad(input_, (0, padw, 0, padh), "reflect")."""
    padh, padw = pad_amounts(x.shape[2], x.shape[3], factor)
    return np.pad(x, ((0, 0), (0, 0), (0, padh), (0, padw)), mode="reflect")


def tile_starts(length: int, crop_size: int, overlap_size: int):
    """:33-36 — range(0, L, crop - overlap), trailing starts whose tile reaches the border dropped,
    then the flush-right start appended."""
    starts = [x for x in range(0, length, crop_size - overlap_size)]
    while starts and starts[-1] + crop_size >= length:
        starts.pop()
    starts.append(length - crop_size)
    return starts


def splitimage(x: np.ndarray, crop_size: int = 128, overlap_size: int = 64):
    """:31-48 — list of [B,C,crop,crop] crops, row starts outer, column starts inner."""
    _, _, H, W = x.shape
    hstarts, wstarts = tile_starts(H, crop_size, overlap_size), tile_starts(W, crop_size, overlap_size)
    starts, split_data = [], []
    for hs in hstarts:
        for ws in wstarts:
            split_data.append(x[:, :, hs:hs + crop_size, ws:ws + crop_size])
            starts.append((hs, ws))
    return split_data, starts


def mergeimage(split_data, starts, crop_size: int = 128, resolution=(1, 3, 128, 128)) -> np.ndarray:
    """:65-75 — accumulate scoremap * tile and scoremap (ones) in list order, divide (float32)."""
    B, C, H, W = resolution
    tot_score = np.zeros((B, C, H, W), np.float32)
    merge_img = np.zeros((B, C, H, W), np.float32)
    scoremap = np.ones((B, C, crop_size, crop_size), np.float32)
    for simg, (hs, ws) in zip(split_data, starts):
        merge_img[:, :, hs:hs + crop_size, ws:ws + crop_size] += scoremap * simg
        tot_score[:, :, hs:hs + crop_size, ws:ws + crop_size] += scoremap
    return merge_img / tot_score


def crop_clamp(x: np.ndarray, h: int, w: int) -> np.ndarray:
    """:172-175 — restored[:, :, :h, :w] clamped to [0, 1]."""
    return np.clip(x[:, :, :h, :w], 0.0, 1.0).astype(np.float32)


def to_uint8(x_nchw: np.ndarray) -> np.ndarray:
    """:176-190 — permute to NHWC and skimage.img_as_ubyte of a float32 image in [0, 1]: multiply
    by 255 in float32, round half to even, clip, cast."""
    img = np.transpose(x_nchw, (0, 2, 3, 1)).astype(np.float32)
    out = np.multiply(img, np.float32(255.0), dtype=np.float32)
    np.rint(out, out=out)
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)
