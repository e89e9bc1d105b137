"""The smallest data layer the test entry point needs: paired image folders read with PIL, yielding the
dict the reference's loaders yield (`lq`, `gt` as RGB float32 CHW in [0, 1], `lq_path`, `gt_path`;
basicsr/data/paired_image_dataset.py:269-319 in its test phase: no crop, no augmentation)."""
from __future__ import annotations

import os

import numpy as np

_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")


def _read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
    return np.ascontiguousarray(a.transpose(2, 0, 1))


class PairedImageFolder:
    def __init__(self, opt):
        self.opt = opt
        self.gt_folder, self.lq_folder = opt["dataroot_gt"], opt["dataroot_lq"]
        tmpl = opt.get("filename_tmpl", "{}")
        self.paths = []
        for name in sorted(os.listdir(self.gt_folder)):
            base, ext = os.path.splitext(name)
            if ext.lower() not in _EXT:
                continue
            lq = os.path.join(self.lq_folder, tmpl.format(base) + ext)
            if not os.path.exists(lq):
                raise FileNotFoundError(f"{lq} (the low-quality partner of {name}) is missing")
            self.paths.append({"lq_path": lq, "gt_path": os.path.join(self.gt_folder, name)})
        mean, std = opt.get("mean"), opt.get("std")
        if mean is not None or std is not None:
            raise NotImplementedError("mean / std normalisation is not provided")

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        p = self.paths[i]
        return {"lq": _read_rgb(p["lq_path"])[None], "gt": _read_rgb(p["gt_path"])[None],
                "lq_path": [p["lq_path"]], "gt_path": [p["gt_path"]]}


class SequentialLoader:
    """batch_size 1, no shuffling: what create_dataloader builds for a test set."""
    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        return (self.dataset[i] for i in range(len(self.dataset)))


def create_dataset(dataset_opt):
    if not str(dataset_opt.get("type", "")).startswith("Dataset_PairedImage"):
        raise ValueError(f"Dataset {dataset_opt.get('type')} is not found.")
    if (dataset_opt.get("io_backend") or {"type": "disk"}).get("type", "disk") != "disk":
        raise NotImplementedError("only the disk io_backend is provided")
    return PairedImageFolder(dataset_opt)


def create_dataloader(dataset, dataset_opt=None, **_ignored):
    return SequentialLoader(dataset)
