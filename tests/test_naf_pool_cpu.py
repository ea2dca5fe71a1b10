"""The wave-pool scheduler of k_multiply_vartime_pool (cheetah_amd/csrc/kernels.hip.h), modelled on the CPU
(tools/naf_pool_sim.py: the kernel's phase policy statement for statement) and run on seeded width-5 NAFs
(src/scalar.rs:585-643): every item takes exactly its 256 doublings and one addition per non-zero digit (an addition phase going on
with the up-to-five doublings that follow it), the loop terminates far inside the kernel's hard phase bound, and the
lane utilisation is what DESIGN.md quotes."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import naf_pool_sim as sim  # noqa: E402


def test_wnaf5_reconstructs_the_scalar_and_is_sparse():
    rng = random.Random(5)
    for _ in range(50):
        k = rng.randrange(sim.Q)
        d = sim.wnaf5(k)
        assert sum(x << i for i, x in enumerate(d)) == k
        assert all(x == 0 or (x % 2 != 0 and -16 < x < 16) for x in d)
        nz = [i for i, x in enumerate(d) if x]
        assert all(b - a >= 5 for a, b in zip(nz, nz[1:]))          # what the packed form (one digit per 4 positions) relies on


def test_every_item_finishes_inside_the_phase_bound():
    rng = random.Random(7)
    for slots in (1, 2, 3):
        nafs = [[sim.wnaf5(rng.randrange(sim.Q)) for _ in range(64)] for _ in range(slots)]
        # edge items: zero, one, the largest canonical scalar, a run of equal scalars (every lane parks together)
        nafs[0][0] = sim.wnaf5(0)
        nafs[0][1] = sim.wnaf5(1)
        nafs[0][2] = sim.wnaf5(sim.Q - 1)
        for l in range(8, 24):
            nafs[slots - 1][l] = nafs[slots - 1][8]
        n_d, n_a, used = sim.run_wave(nafs, slots)               # asserts termination, the bound and completion itself
        steps = sum(256 + sum(1 for x in naf if x) for row in nafs for naf in row)
        assert used == steps                                      # no step lost, none repeated
        assert n_d + n_a < sim.phase_bound(slots) // 20


def test_two_slots_keep_the_lanes_busy():
    r1 = sim.simulate(1, 4, seed=11)
    r2 = sim.simulate(2, 4, seed=11)
    assert r2["lane_utilisation"] > 0.96 and r2["efficiency"] > 0.96
    assert r1["lane_utilisation"] < 0.85                          # one item per lane: what the pool is for
    # the doublings that follow an addition ride in its phase: ~ 100 phases per 64 items (one load and one store of the
    # accumulator each) instead of the ~ 300 of one step per phase, at nearly the same lane utilisation
    phases = (r2["doubling_phases_per_wave"] + r2["addition_phases_per_wave"]) / 2
    assert 80 < phases < 115
    r0 = sim.simulate(2, 4, seed=11, run=0)
    assert abs(r0["doubling_phases_per_wave"] / 2 - 256) < 6 and r0["lane_utilisation"] > 0.98
    assert r2["time_per_64_items"] < 1.04 * r0["time_per_64_items"]


def test_fused_runs_never_skip_a_digit():
    """run_wave asserts, for every doubling of a fused run but the last, that the digit it arrives at is zero: with
    runs of 5 that is the width-5 NAF's guarantee of four zeros after a non-zero digit -- a run of 6 must trip it."""
    import pytest
    rng = random.Random(13)
    nafs = [[sim.wnaf5(rng.randrange(sim.Q)) for _ in range(64)] for _ in range(2)]
    sim.run_wave(nafs, 2, run=5)
    with pytest.raises(AssertionError):
        sim.run_wave(nafs, 2, run=7)
