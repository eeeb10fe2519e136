"""The golden recipe itself is under test: `python tests/golden/make_golden.py` -- unflagged, as its docstring documents --
must run end to end against the live reference and reproduce every committed fixture bit for bit (the
recipe crashed half way on numpy >= 2.0 because it deleted numpy's own np.bool).  Needs /root/reference, i.e. the build
container; skipped on the GPU box."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/alegnn"), reason="the reference is only in the build container")


def _same(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind in "fc":
        return np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


def test_unflagged_recipe_reproduces_every_committed_fixture(tmp_path):
    out = str(tmp_path / "gold")
    r = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py"), "--out", out], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    committed = sorted(glob.glob(os.path.join(GOLD, "*.npz")))
    assert len(committed) >= 58
    for f in committed:
        g = os.path.join(out, os.path.basename(f))
        assert os.path.exists(g), f"the unflagged recipe did not write {os.path.basename(f)}"
        a, b = np.load(f, allow_pickle=True), np.load(g, allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), os.path.basename(f)
        for k in a.files:
            assert _same(a[k], b[k]), f"{os.path.basename(f)}[{k}] differs from the committed fixture"
    # and nothing is produced that is not committed (a fixture nobody checks in is a fixture nobody pins)
    extra = {os.path.basename(p) for p in glob.glob(os.path.join(out, "*.npz"))} - {os.path.basename(p) for p in committed}
    assert not extra, extra
    # numpy's own np.bool survives the run of the coarsening cases (the defect itself)
    r2 = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['x','--coarsen-only','--out',%r]; import runpy, numpy as np; "
                         "runpy.run_path(%r, run_name='__main__'); assert np.bool is np.bool_" % (out, os.path.join(GOLD, "make_golden.py"))],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
