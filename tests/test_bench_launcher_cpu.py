"""`python bench.py --gpus N` must start its own rank processes for N > 1 (the form the
driver runs).  On a box without GPUs the same launcher, sharding plan, chunked all-gather
and JSON plumbing are driven over gloo with a stand-in for the arithmetic
(`--rehearse-cpu`); the RCCL leg differs only in the backend and the device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, BENCH] + list(extra), capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_starts_two_ranks_and_forwards_one_line():
    r = _run("--gpus", "2", "--rehearse-cpu", "--log2n", "12", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["rehearsal"] is True and line["gathered_ok"] is True
    assert line["config"]["per_gpu"] == 2048 and line["config"]["allgather_chunks"] == 8
    assert line["scaling"] == "strong" and line["unit"] == "scalar_mults/s" and line["value"] > 0
    # both modes are reported for every N: with the all-gather (value) and without it (no_gather_value)
    assert line["compute_ms"] > 0 and line["no_gather_value"] > 0 and "gather_ms_exposed" in line
    # the gathered array is the same whatever the number of ranks
    one = json.loads(_run("--rehearse-cpu", "--log2n", "12", "--steps", "1", "--warmup", "0").stdout.strip())
    assert one["n_gpus"] == 1 and one["checksum"] == line["checksum"]
    assert one["ranks_seen"] == 1 and one["compute_ms"] > 0 and one["no_gather_value"] > 0 and "gather_ms_exposed" in one


def test_launcher_fails_clearly_without_enough_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    r = _run("--gpus", "2", "--log2n", "12")
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()


def test_a_failing_rank_fails_the_launcher():
    # n not divisible by world * chunks: every rank raises, the launcher must not print a line
    r = _run("--gpus", "2", "--rehearse-cpu", "--log2n", "3", "--chunks", "8")
    assert r.returncode != 0 and not r.stdout.strip()
