"""bench.py's multi-GPU code path on the one-GPU box: --rccl-at-one makes a one-rank NCCL (= RCCL) process group with the
options the N > 1 run asks for and sends every chunk's results through the in-place all_gather_into_tensor from the side
stream, plus the all-reduces behind ranks_seen / max-over-ranks.  The gathered result must be the plain run's (same
checksum).  Execution evidence for the path the driver runs at N = 2, 4, 8; the scaling itself can only be measured there."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--log2n", "20", "--steps", "1", "--warmup", "0", "--no-secondary",
           "--no-cpu-baseline"] + list(flags)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_rccl_path_in_a_one_rank_group_gives_the_same_results():
    plain = _bench()
    coll = _bench("--rccl-at-one")
    assert coll["rccl_at_one"] is True and coll["ranks_seen"] == 1 and coll["config"]["allgather_chunks"] == 8
    assert plain["config"]["allgather_chunks"] == 1
    assert coll["checksum"] == plain["checksum"] and coll["identities"] == plain["identities"]
    assert "compute_ms" in coll and "gather_ms_exposed" in coll
