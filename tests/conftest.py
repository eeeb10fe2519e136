import os
import sys

import pytest

# the tuning knobs (gf_tune) are refused unless the process opted in BEFORE libgfhip.so is loaded; the tests use them to force
# each pipeline / kernel variant.  Default knob values == the product's, so everything not explicitly tuned runs the product path.
os.environ.setdefault("GFHIP_EXPERIMENTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "graph-neural-networks_amd")
for p in (ROOT, PKG_ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests must never silently pass on a box without a GPU: skip them there
    unless -m gpu was asked for explicitly (then they fail loudly in the product code)."""
    import torch

    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
