"""Image plumbing of the reference's test scripts on the device (same function names and argument
meaning as Deraining/test_rain100L.py:31-75,147-190), through the C ABI of libjirf_hip.  torch is
used for device memory only; there is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import check, lib

_FP = C.POINTER(C.c_float)


def _ptr(t: torch.Tensor):
    return C.cast(t.data_ptr(), _FP)


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev(x: torch.Tensor) -> torch.Tensor:
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise TypeError("expected a float32 NCHW tensor on the GPU")
    return x.contiguous()


def pad_amounts(h: int, w: int, factor: int = 8) -> Tuple[int, int]:
    """test_rain100L.py:148-151."""
    H, W = ((h + factor) // factor) * factor, ((w + factor) // factor) * factor
    return (H - h if h % factor != 0 else 0), (W - w if w % factor != 0 else 0)


def reflect_pad(x: torch.Tensor, pad_bottom: int, pad_right: int) -> torch.Tensor:
    """nn.pad(x, (0, pad_right, 0, pad_bottom), "reflect") (test_rain100L.py:152;
    basicsr/models/image_restoration_model.py:286)."""
    x = _dev(x)
    b, c, h, w = x.shape
    if pad_bottom == 0 and pad_right == 0:
        return x
    y = torch.empty((b, c, h + pad_bottom, w + pad_right), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().jirf_op_reflect_pad(_ptr(x), _ptr(y), b, c, h, w, pad_bottom, pad_right, _stream(x)))
    return y


def pad_to_factor(x: torch.Tensor, factor: int = 8) -> torch.Tensor:
    """test_rain100L.py:147-152: reflect-pad bottom / right to a multiple of `factor`."""
    _, _, h, w = x.shape
    return reflect_pad(x, *pad_amounts(h, w, factor))


def square_side(h: int, w: int, multiple: int = 128) -> int:
    """Side of the smallest square with side % multiple == 0 that holds an h x w image."""
    return -(-max(h, w) // multiple) * multiple


def pad_to_square(x: torch.Tensor, multiple: int = 128) -> torch.Tensor:
    """Reflect-pad bottom / right to the enclosing square (side % multiple == 0).  The reference's AST
    reshapes its tokens to sqrt(L) x sqrt(L) with windows of 8 at four scales (ast_arch.py:573, 592, 668,
    770), so it only takes such squares; its scripts never feed it anything else.  Padding other images up
    to one is this package's choice, not behaviour of the reference."""
    _, _, h, w = x.shape
    side = square_side(h, w, multiple)
    if side - h >= h or side - w >= w:
        raise ValueError(f"reflect padding {h}x{w} to {side}x{side} needs more rows or columns than the image has")
    return reflect_pad(x, side - h, side - w)


def tile_starts(length: int, crop_size: int, overlap_size: int) -> List[int]:
    """test_rain100L.py:33-36."""
    if crop_size <= overlap_size:
        raise ValueError("crop_size must exceed overlap_size")
    if crop_size > length:
        raise ValueError(f"tile {crop_size} does not fit a side of {length}")
    starts = list(range(0, length, crop_size - overlap_size))
    while starts and starts[-1] + crop_size >= length:
        starts.pop()
    starts.append(length - crop_size)
    return starts


def _grid(starts: Sequence[Tuple[int, int]]):
    hs = sorted({s[0] for s in starts}, key=[s[0] for s in starts].index)
    ws = sorted({s[1] for s in starts}, key=[s[1] for s in starts].index)
    if [(a, b) for a in hs for b in ws] != [tuple(s) for s in starts]:
        raise ValueError("starts must be the row-major product of row starts and column starts")
    return np.asarray(hs, np.int32), np.asarray(ws, np.int32)


def splitimage(imgtensor: torch.Tensor, crop_size: int = 128, overlap_size: int = 64):
    """test_rain100L.py:31-48.  Returns (split_data, starts); split_data is a list of
    [B,C,crop,crop] views of ONE device tensor [n_tiles*B, C, crop, crop] (attribute `.stacked` of
    the list) so the tiles can go through the network as a batch."""
    x = _dev(imgtensor)
    b, c, h, w = x.shape
    hs = np.asarray(tile_starts(h, crop_size, overlap_size), np.int32)
    ws = np.asarray(tile_starts(w, crop_size, overlap_size), np.int32)
    n = len(hs) * len(ws)
    tiles = torch.empty((n * b, c, crop_size, crop_size), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().jirf_op_split_tiles(_ptr(x), _ptr(tiles), b, c, h, w, crop_size, hs.ctypes.data, len(hs),
                                        ws.ctypes.data, len(ws), _stream(x)))
    split = TileList(tiles[i * b:(i + 1) * b] for i in range(n))
    split.stacked = tiles
    return split, [(int(a), int(d)) for a in hs for d in ws]


class TileList(list):
    stacked: torch.Tensor = None


def mergeimage(split_data, starts, crop_size: int = 128, resolution=(1, 3, 128, 128)) -> torch.Tensor:
    """test_rain100L.py:65-75 with the uniform score map the scripts use: mean blend in list order."""
    b, c, h, w = (int(v) for v in resolution)
    hs, ws = _grid(starts)
    stacked = getattr(split_data, "stacked", None)
    if stacked is None or stacked.shape[0] != len(starts) * b:
        stacked = torch.cat([_dev(t) for t in split_data], dim=0)
    stacked = _dev(stacked)
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=stacked.device)
    with torch.cuda.device(stacked.device):
        check(lib().jirf_op_merge_tiles(_ptr(stacked), _ptr(out), b, c, h, w, crop_size, hs.ctypes.data, len(hs),
                                        ws.ctypes.data, len(ws), _stream(stacked)))
    return out


def crop_clamp(restored: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """test_rain100L.py:171-175: unpad and clamp to [0, 1]."""
    x = _dev(restored)
    b, c, H, W = x.shape
    y = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().jirf_op_crop_clamp(_ptr(x), _ptr(y), b, c, H, W, h, w, _stream(x)))
    return y


def to_uint8_image(restored: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """test_rain100L.py:171-190: unpad, clamp, NHWC, img_as_ubyte.  Returns uint8 [B,h,w,C] on the GPU."""
    x = _dev(restored)
    b, c, H, W = x.shape
    y = torch.empty((b, h, w, c), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().jirf_op_crop_clamp_u8(_ptr(x), C.c_void_p(y.data_ptr()), b, c, H, W, h, w, _stream(x)))
    return y


def restore_image(net, img: torch.Tensor, factor: int = 8, crop_size: int = None, overlap_size: int = None,
                  tile_batch: int = 32, square_multiple: int = None, carry_cache: bool = False) -> torch.Tensor:
    """One image through a test script's loop body (test_rain100L.py:147-190): pad, restore (whole
    image, or `crop_size` tiles with `overlap_size` overlap as the FPro / GoPro branches do),
    unpad, clamp.  img: [B,3,h,w] float32 on the GPU; returns [B,3,h,w].  With `square_multiple` the image
    is padded to the enclosing square instead (`pad_to_square`, for AST).

    Tiles are independent B = 1 calls in the scripts, so tile batches always run with per-image batch
    semantics (no statistic of one tile reaches another), whatever `net.batch_mode` says, and the result
    does not depend on `tile_batch`.  For a network that carries state between calls (HINT's `qv_cache`,
    hint_arch.py:511,660-696) the tiled branch has two modes (the whole-image branch is one plain call
    and leaves the state to the network, which carries it like the reference module):
      carry_cache=False  every tile starts cold (`qv_cache = None`), so tiles can be batched and sharded;
                         the cache is None again on return;
      carry_cache=True   the script's literal loop (Motion_Deblurring/test_GoPro.py:142-171): tiles go
                         through one at a time in list order and each starts from the cache the previous
                         call left (the first from whatever the network holds on entry); the last tile's
                         cache stays in the network.  Needs a single image (B = 1)."""
    x = _dev(img)
    h, w = x.shape[2], x.shape[3]
    xp = pad_to_square(x, square_multiple) if square_multiple else pad_to_factor(x, factor)
    stateful = hasattr(net, "qv_cache")
    if carry_cache and not stateful:
        raise ValueError("carry_cache needs a network with a qv_cache (HINT)")
    if crop_size is None:
        restored = net(xp)      # one call: the network's own call-to-call state applies, as in the scripts
    else:
        split, starts = splitimage(xp, crop_size, overlap_size)
        tiles = split.stacked
        had_mode = getattr(net, "batch_mode", None)
        try:
            if carry_cache:
                if x.shape[0] != 1:
                    raise ValueError("carry_cache replays the scripts' B = 1 loop: pass one image")
                if had_mode is not None:
                    net.batch_mode = "coupled"          # a literal B = 1 call; the cache stays [1,C,C]
                outs = [net(tiles[i:i + 1]) for i in range(tiles.shape[0])]
            else:
                if had_mode is not None:
                    net.batch_mode = "per_image"
                outs = []
                for i in range(0, tiles.shape[0], tile_batch):
                    if stateful:
                        net.qv_cache = None
                    outs.append(net(tiles[i:i + tile_batch]))
                if stateful:
                    net.qv_cache = None
        finally:
            if had_mode is not None:
                net.batch_mode = had_mode
        done = TileList()
        done.stacked = torch.cat(outs, dim=0)
        restored = mergeimage(done, starts, crop_size, tuple(xp.shape))
    return crop_clamp(restored, h, w)


def restore_images(net, images: Sequence[torch.Tensor], batch: int = 8, balance: bool = False, gather: bool = True,
                   dst: int = 0, group=None, **restore_kwargs):
    """The scripts' loop over a folder of images (Deraining/test_rain100L.py:141-190), spread over the ranks of a
    process group: every rank passes the same list of [3,h,w] / [1,3,h,w] float32 images (host or device), image i
    is restored by rank i % ranks (balance=True: by pixel count, for folders of mixed sizes) on that rank's GPU
    through `restore_image(net, ..., **restore_kwargs)`, and rank `dst` receives the restored images in list order
    (the other ranks None; gather=False: every rank keeps its {index: image}).  Consecutive images of one shape on a
    rank go through the network `batch` at a time with per-image batch semantics - each is still its own B = 1
    call of the reference - and a network that carries state between calls (HINT's qv_cache) starts every call cold,
    so the result does not depend on the number of ranks or on `batch`.  Weights are replicated; nothing but the
    final gather crosses ranks."""
    from . import sharding
    dev = getattr(net, "_device", None)
    device = torch.device("cuda", torch.cuda.current_device() if dev is None else dev)
    stateful = hasattr(net, "qv_cache")

    def shape_of(t):
        return tuple(t.shape[-3:])

    def run(indices: List[int], mine: list) -> list:
        out: List[torch.Tensor] = []
        had_mode = getattr(net, "batch_mode", None)
        try:
            if had_mode is not None:
                net.batch_mode = "per_image"
            i = 0
            while i < len(mine):
                j = i + 1
                while j < len(mine) and j - i < batch and shape_of(mine[j]) == shape_of(mine[i]):
                    j += 1
                x = torch.stack([t.reshape(shape_of(t)) for t in mine[i:j]]).to(device, torch.float32)
                if stateful:
                    net.qv_cache = None
                y = restore_image(net, x, **restore_kwargs)
                out.extend(y[k] for k in range(j - i))
                i = j
            if stateful:
                net.qv_cache = None
        finally:
            if had_mode is not None:
                net.batch_mode = had_mode
        return out

    costs = [float(t.shape[-1] * t.shape[-2]) for t in images] if balance else None
    return sharding.run_sharded(list(images), run, costs=costs, gather=gather, dst=dst, group=group)
