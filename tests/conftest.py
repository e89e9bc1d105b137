import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-restoration_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_report_header(config):
    """The Jittor-semantic assumptions the oracle runs under (oracle/common.py JittorAssumptions plus the
    per-network switches), printed once per session so that a report states what was assumed."""
    from oracle.common import ASSUMPTIONS
    from oracle.fpro_oracle import FproSwitches
    from oracle.hint_oracle import HintSwitches
    lines = ["oracle assumptions (Jittor semantics, unverifiable offline): " + ", ".join(ASSUMPTIONS.describe())]
    for sw in (HintSwitches(), FproSwitches()):
        lines.append(f"  {type(sw).__name__}: " + ", ".join(f"{k}={v!r}" for k, v in vars(sw).items()))
    return lines
