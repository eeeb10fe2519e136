"""Short, fixed-seed runs of the fuzz tools (tools/*_fuzz.py: random shapes through the host layer against the CPU oracles / SELL-8); the long runs are
recorded under profiles/r06_j_fuzz, r06_k_layerfuzz, r06_n_evgf_fuzz, r06_o_nvgf_db_fuzz.  Reference lines: graphML.py:83-176 (LSIGF), :389-488 (EVGF),
:490-600 (NVGF), :1096-1290 (LSIGF_DB), :3395-3538 (GRNN_DB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tool, *args, env=None, timeout=600):
    e = dict(os.environ)
    e["GFHIP_EXPERIMENTS"] = "0"         # the configuration a user's process has: shipped heuristics only (the oracle fuzz tools never call gf_tune)
    e.pop("GFHIP_LIB", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_layers_on_large_graphs_against_the_oracle():
    out = run("layer_fuzz.py", 6, 1)
    assert out.count("\nok ") + out.startswith("ok ") == 6


def test_layers_on_small_graphs_and_at_the_dispatch_boundaries_against_the_oracle():
    run("layer_fuzz.py", 25, 22, env=dict(FUZZ_N="37,64,100,333,1000,1279,1280,1682,2559,2560,5000,5119,5120,8000,12000,16000,30000,48000,49152", FUZZ_B="1,2,4,7,8,16,20,32,100"))


def test_edge_variant_filters_against_the_oracle():
    run("evgf_fuzz.py", 30, 1)


def test_node_variant_and_time_varying_filters_against_their_oracles():
    run("nvgf_db_fuzz.py", 60, 2)


def test_khop_chains_bitwise_against_sell8():
    run("msweep_fuzz.py", 10, 1, env=dict(GFHIP_EXPERIMENTS="1"))
