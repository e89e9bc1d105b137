"""`python -m basicsr.test -opt <options.yml>` — the reference's test entry point (basicsr/test.py:13-60):
read the option file, build the test sets and the model, run validation on each set."""
from __future__ import annotations

import argparse
import logging

import yaml

from basicsr.data import create_dataloader, create_dataset
from basicsr.models import create_model


def parse_options(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    with open(args.opt, "r") as f:
        opt = yaml.safe_load(f)
    opt["is_train"] = False
    opt["dist"] = False
    opt["device"] = args.device
    for phase, dataset in (opt.get("datasets") or {}).items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
    return opt


def main(argv=None):
    opt = parse_options(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s: %(message)s")
    logger = logging.getLogger("basicsr")
    test_loaders = []
    for _, dataset_opt in sorted((opt.get("datasets") or {}).items()):
        test_set = create_dataset(dataset_opt)
        logger.info("Number of test images in %s: %d", dataset_opt.get("name"), len(test_set))
        test_loaders.append(create_dataloader(test_set, dataset_opt))
    model = create_model(opt)
    results = {}
    for loader in test_loaders:
        name = loader.dataset.opt.get("name", "val")
        logger.info("Testing %s...", name)
        val = opt.get("val") or {}
        model.validation(loader, current_iter=opt.get("name"), tb_logger=None, save_img=val.get("save_img", False),
                         rgb2bgr=val.get("rgb2bgr", True), use_image=val.get("use_image", True))
        results[name] = dict(model.metric_results)
    return results


if __name__ == "__main__":
    main()
