#!/usr/bin/env python3
"""CPU model of the wave-pool scheduler of k_multiply_vartime_pool (csrc/kernels.hip.h): one wave of 64 lanes owns
SLOTS x 64 scalar multiplications, each following the reference's own NAF sequence (256 doublings, one addition after
every non-zero width-5 NAF digit; src/curve/projective.rs:633-662, src/scalar.rs:585-643).  A phase takes up to 64 items
that all want the same step: one doubling, or an addition TOGETHER WITH the RUN = 5 doublings that follow it in every
item's sequence (a non-zero width-5 NAF digit is followed by four zeros, so the four doublings after an addition never
meet a digit and the fifth is the first that can), for one load and one store of the accumulator.  The model is the kernel's policy statement for statement (choose_phase / take below are
what the kernel computes with ballots and popcounts); tests/test_naf_pool_cpu.py runs it on seeded scalars to show that
every item finishes, within the kernel's hard phase bound, and what the lane utilisation is.

    python tools/naf_pool_sim.py [slots] [waves]
"""
import random
import sys

Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
DBL_COST, ADD_COST = 1.0, 1.65     # issue time of a doubling phase and of an addition phase (3 060 : 5 055 instructions)
RUN = 5                            # doublings fused behind an addition (CH_VT_RUN in the kernel; 0 = none)
RUN_SWITCH = 60                    # one fewer once the candidates' mean position is at or below this (CH_VT_RUN_SWITCH)


def wnaf5(k):
    """width-5 NAF, one digit per bit position, as bytes_to_wnaf_vartime(by, 5) for k < 2^255"""
    d = [0] * 257
    pos = 0
    while k:
        if k & 1:
            z = k & 31
            if z >= 16:
                z -= 32
            d[pos] = z
            k -= z
        k >>= 1
        pos += 1
    return d[:256]


def phase_bound(slots):
    """the kernel's hard bound on phases per round: every phase completes at least one step of one item, and an item
    has 256 doublings and at most 52 additions (a non-zero digit is followed by four zeros)"""
    return slots * 64 * (256 + 52)


def choose_phase(n_dbl, n_add):
    """-> 'a', 'd' or None.  Full phases first (additions before doublings: they free parked items), else the larger
    group."""
    if n_add >= 64:
        return "a"
    if n_dbl >= 64:
        return "d"
    if n_add == 0 and n_dbl == 0:
        return None
    return "a" if n_add > n_dbl else "d"


def take(want, pos, slots, phase):
    """the up-to-64 items of a phase: the candidates at or above the candidates' mean position first (the laggards:
    positions count down), then the others; inside a tier the slot order and the lane order rotate with the phase
    number so that no slot and no lane is always last"""
    cand = [(s, l) for s in range(slots) for l in range(64) if want[s][l]]
    cnt = len(cand)
    total = sum(pos[s][l] for s, l in cand)
    r = (phase * 17) & 63
    order = [(o + phase) % slots for o in range(slots)]
    items = []
    for tier in (True, False):
        for s in order:
            for j in range(64):
                l = (j + r) & 63
                if want[s][l] and ((pos[s][l] * cnt >= total) == tier):
                    items.append((s, l))
    return items[:64]


def run_wave(nafs, slots, run=None, stats=None):
    """nafs[s][l]: digit lists.  Returns (doubling phases, addition phases, steps executed); stats, if given, receives
    the lane-step slots issued (64 x the steps of every phase) and the modelled issue time."""
    run = RUN if run is None else run
    pos = [[255] * 64 for _ in range(slots)]
    pd = [[0] * 64 for _ in range(slots)]
    n_d = n_a = used = 0
    issued = 0
    time = 0.0
    phase = 0
    bound = phase_bound(slots)
    while True:
        want_d = [[pd[s][l] == 0 and pos[s][l] >= 0 for l in range(64)] for s in range(slots)]
        want_a = [[pd[s][l] != 0 for l in range(64)] for s in range(slots)]
        kind = choose_phase(sum(map(sum, want_d)), sum(map(sum, want_a)))
        if kind is None:
            break
        assert phase < bound, "phase bound exceeded"
        want = want_a if kind == "a" else want_d
        items = take(want, pos, slots, phase)
        assert items
        cnt = sum(map(sum, want))
        total = sum(pos[s][l] for s in range(slots) for l in range(64) if want[s][l])
        run_now = 0 if run == 0 else (run if total > RUN_SWITCH * cnt else run - 1)
        longest = 0
        for s, l in items:
            ndbl = 1
            if kind == "a":
                pd[s][l] = 0
                used += 1
                ndbl = min(run_now, pos[s][l] + 1)
            for j in range(ndbl):
                assert pd[s][l] == 0          # only the last doubling of a run can meet a non-zero digit
                pd[s][l] = nafs[s][l][pos[s][l]]
                pos[s][l] -= 1
            used += ndbl
            longest = max(longest, ndbl)
        if kind == "a":
            n_a += 1
            issued += 64 * (1 + longest)
            time += ADD_COST + longest * DBL_COST
        else:
            n_d += 1
            issued += 64
            time += DBL_COST
        phase += 1
    assert all(p == -1 for row in pos for p in row) and all(d == 0 for row in pd for d in row)
    if stats is not None:
        stats["issued"] = stats.get("issued", 0) + issued
        stats["time"] = stats.get("time", 0.0) + time
    return n_d, n_a, used


def simulate(slots, waves, seed=1, run=None):
    rng = random.Random(seed)
    tot_d = tot_a = tot_used = 0
    ideal = 0.0
    stats = {}
    for _ in range(waves):
        nafs = [[wnaf5(rng.randrange(Q)) for _ in range(64)] for _ in range(slots)]
        n_d, n_a, used = run_wave(nafs, slots, run, stats)
        tot_d += n_d
        tot_a += n_a
        tot_used += used
        ideal += sum(256 * DBL_COST + sum(1 for x in naf if x) * ADD_COST for row in nafs for naf in row) / 64
    return {"slots": slots, "waves": waves, "run": RUN if run is None else run,
            "doubling_phases_per_wave": tot_d / waves, "addition_phases_per_wave": tot_a / waves,
            "lane_utilisation": tot_used / float(stats["issued"]), "time_per_64_items": stats["time"] / waves / slots,
            "ideal_time_per_64_items": ideal / waves / slots, "efficiency": ideal / stats["time"]}


if __name__ == "__main__":
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    run = int(sys.argv[3]) if len(sys.argv) > 3 else None
    print(simulate(slots, waves, run=run))
