"""The randomised differential run of tests/fuzz_gpu.py (every entry point, random forms and batch sizes, special and
off-curve points, special scalars; GPU against the oracle, bit-exact) as a bounded, deterministic part of the GPU suite:
a fixed number of runs per seed, so a failure names a (seed, run) pair that fuzz_gpu.py reproduces."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 20261111])
def test_bounded_fuzz_against_the_oracle(seed):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import fuzz_gpu
    runs, findings = fuzz_gpu.fuzz(budget=100.0, seed=seed, max_n=900, max_runs=1500)
    assert not findings, "\n".join(findings)
    assert runs >= 300, "the fuzz ran %d cases only: the oracle side has become too slow for its budget" % runs
