"""The constant-time claim, re-checked on the ISA of the library that is in the tree (tools/check_ct_isa.py):
no conditional branch and no memory address in the window loops of the constant-time kernels depends on a digit;
the variable-time kernels must be flagged (the analysis is not blind).  Reference behaviour: the constant-time scan
of LookupTable::get_point, src/lookup.rs:81-102.  CPU only: disassembles the gfx950 code object, runs nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constant_time_kernels_have_no_digit_dependent_branch_or_address():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ct_isa.py")], capture_output=True, text=True, timeout=900)
    rep = json.loads(r.stdout)
    bad = {k: v for k, v in rep["kernels"].items() if not v["ok"]}
    assert r.returncode == 0 and rep["ok"] and not bad, json.dumps(bad, indent=1)[:4000]
    kinds = [v["expect"] for v in rep["kernels"].values()]
    assert kinds.count("ct") >= 19 and kinds.count("vt_address") >= 6 and kinds.count("vt_branch") >= 7
    assert all(v["digit_sources"] > 0 and v["window_loops"] > 0 for v in rep["kernels"].values() if v["expect"] == "ct")
