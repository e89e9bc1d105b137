"""Synthetic weights and images (kept under this name for the tests): the generators live with the
product's host code in jirf_hip/synthetic.py, because the benchmark and the smoke run use them too
and nothing outside the tests may depend on oracle/."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-restoration_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from jirf_hip.synthetic import (make_ast_state, make_astv2_state, make_fpro_state, make_hint_state,  # noqa: E402,F401
                                synthetic_images)
