#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): variable-base constant-time Cheetah scalar
multiplications per second, plus Fp6 multiplications per second.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 5): 2^26 seeded (P_i, k_i) pairs, out[i] = k_i * P_i by
AffinePoint::multiply (src/curve/affine.rs:373-375 -> JacobianPoint::multiply,
src/curve/jacobian.rs:599-611), affine results.  The
2^26 pairs are sharded over the N ranks (strong scaling: the total is fixed) and
the affine results are all-gathered over RCCL so that every rank holds all of
them.  Inputs are synthetic, generated on the device (points r_i * G, seeded
scalars) before the timed region; a step is one full pass.  One JSON line on
stdout from rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# reference-algorithm work per unit (SURVEY.md 8d / BASELINE.md 2; derivation in DESIGN.md 5).
# Affine in / affine out is AffinePoint::multiply, which runs JacobianPoint::multiply
# (src/curve/affine.rs:373-375, src/curve/jacobian.rs:599-611):
#   LookupTable<16>::from(&JacobianPoint): normalise 300 + 15 complete adds x 300 + batch_normalize(16) 2 103 = 6 903
#   64 windows x (4 x 180 + 297) = 65 088;  - 2^256 S correction 297;  final normalisation 300
FPMUL_PER_CT_MULT = 72588
MAD32_PER_CT_MULT = 4 * FPMUL_PER_CT_MULT   # 1 Fp mul = 4 v_mad_u64_u32 on 32-bit limbs
MAD32_PER_CT_MULT_PROJECTIVE = 348468       # ProjectivePoint::multiply, for comparison only
FP6_MUL_BYTES = 144             # 2 x 48 in + 48 out
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=26, help="total scalar multiplications = 2^log2n (26 = BASELINE config 5)")
    ap.add_argument("--chunks", type=int, default=0, help="all-gather chunks per rank (0: 1 at N=1, 8 otherwise)")
    ap.add_argument("--cpu-sample-log2", type=int, default=14, help="CPU baseline sample = 2^k multiplications per core")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reserve-cus", type=int, default=8, help="CUs left free for the RCCL kernel when --gpus > 1")
    return ap.parse_args()


def cpu_baseline(n_sample, seed=5):
    """The oracle (C restatement of the reference CPU path, OpenMP over independent
    elements) timed on this box's host cores on a bounded sample of the same workload."""
    from cheetah_amd import synth
    from oracle import binding as ora
    ora.build()
    cores = ora.lib().ora_num_threads()
    k = synth.scalars(seed, n_sample)
    r = synth.scalars(seed, n_sample, 1 << 26)
    pts, _ = ora.mul_by_generator_batch(r, 0)
    ora.multiply_batch(pts[:cores * 4], 0, k[:cores * 4], 0)  # warm-up, table init
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        ora.multiply_batch(pts, 0, k, 0)
        best = min(best, time.perf_counter() - t0)
    return {"value": n_sample / best, "unit": "scalar_mults/s", "cores": int(cores), "kind": "port",
            "sample": "2^%d of the 2^26 (P,k) pairs, oracle ora_multiply_batch (C restatement of AffinePoint::multiply, "
                      "OpenMP static split over %d threads), best of 2" % (int(np.log2(n_sample)), cores)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(device))

    from cheetah_amd import AFFINE, Engine
    from cheetah_amd.distributed import shard_plan, sharded_apply

    eng = Engine(local_rank)
    if world > 1:
        # the all-gather of chunk c runs while chunk c + 1 computes: keep a few CUs' worth of workgroup slots free so
        # that RCCL's kernel is placed at once instead of waiting for (and then delaying) a persistent compute launch
        eng.set_reserved_cus(args.reserve_cus)
    n = 1 << args.log2n
    chunks = args.chunks or (1 if world == 1 else 8)
    s, plan = shard_plan(n, world, chunks)

    # ---- inputs, resident in HBM before the timed region ---------------------
    # rank-local storage only for the blocks this rank computes: [chunks * s] rows
    pts = torch.empty((chunks * s, 12), dtype=torch.int64, device=device)
    ks = torch.empty((chunks * s, 32), dtype=torch.uint8, device=device)
    tmp_inf = torch.empty((s,), dtype=torch.uint8, device=device)
    for j, (c, g0) in enumerate(plan[rank]):
        r = eng.synth_scalars(5, s, (1 << 26) + g0)          # r_i, stream offset 2^26 of seed 5
        eng.mul_by_generator_batch(r, AFFINE, out=pts[j * s:(j + 1) * s], inf=tmp_inf)  # P_i = r_i * G
        ks[j * s:(j + 1) * s] = eng.synth_scalars(5, s, g0)   # k_i
        del r
    torch.cuda.synchronize()
    local_of = {g0: j for j, (c, g0) in enumerate(plan[rank])}
    def step(timed_kernel=False):
        if timed_kernel:  # one un-overlapped pass, HIP events on the launch stream around each call's kernel launches
            total = 0.0
            launches = 0
            xy = torch.empty((s, 12), dtype=torch.int64, device=device)
            inf = torch.empty((s,), dtype=torch.uint8, device=device)
            upl = eng.multiply_units_per_launch(s, False, AFFINE)  # the library bounds each k_multiply_ct launch
            for j in range(chunks):
                eng._use_torch_stream()
                eng.timer_start()
                eng.multiply_batch(pts[j * s:(j + 1) * s], ks[j * s:(j + 1) * s], AFFINE, AFFINE, out=xy, inf=inf)
                total += eng.timer_stop_ms()
                launches += -(-s // upl)
            return total / launches, chunks * s / launches, launches
        return sharded_apply(lambda g0, cnt, oxy, oinf: eng.multiply_batch(pts[local_of[g0] * s:(local_of[g0] + 1) * s],
                                                                           ks[local_of[g0] * s:(local_of[g0] + 1) * s],
                                                                           AFFINE, AFFINE, out=oxy, inf=oinf),
                             n, 12, rank, world, chunks, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
        del out
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    xy, inf = out
    checksum = int(xy.sum().item()) & 0xFFFFFFFFFFFFFFFF  # wrap-around sum of canonical limbs, identical on every N
    n_inf = int(inf.sum().item())

    # ---- roofline of the dominant kernel (k_multiply_ct), measured live ---------
    ms_launch, units_per_launch, n_launches = step(timed_kernel=True)   # averages over the pass's kernel launches
    mad_peak = eng.calibrate_mad_peak()          # T v_mad_u64_u32 / s, measured on this device
    achieved = units_per_launch * MAD32_PER_CT_MULT / (ms_launch * 1e-3) / 1e12
    traffic = None                               # HBM bytes per launch from the committed PMC passes (profiles/)
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            tr = json.load(f)["k_multiply_ct"]
        traffic = (tr["fetch_bytes_per_unit"] + tr["write_bytes_per_unit"]) * units_per_launch
    except (OSError, KeyError, ValueError):
        pass

    # ---- secondary metric: elementwise Fp6 multiplication (config 1, 2^20 pairs) --
    nf = 1 << 20
    fa, fb = eng.synth_fp6(1, nf, 0), eng.synth_fp6(1, nf, nf)
    for _ in range(3):
        eng.fp6_mul_batch(fa, fb)
    torch.cuda.synchronize()
    eng._use_torch_stream()
    reps = 20
    eng.timer_start()
    for _ in range(reps):
        eng.fp6_mul_batch(fa, fb)
    fp6_ms = eng.timer_stop_ms() / reps
    fp6_rate = nf / (fp6_ms * 1e-3)
    fp6_gbs = nf * FP6_MUL_BYTES / (fp6_ms * 1e-3) / 1e9

    if rank == 0:
        total = n * args.steps
        line = {
            "metric": "Cheetah variable-base scalar mults/sec (1 GPU, node) + Fp6 muls/sec",
            "value": total / dt,
            "unit": "scalar_mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "2^%d variable-base constant-time scalar mults (AffinePoint::multiply = JacobianPoint::multiply ladder), affine in/out, "
                                   "sharded over %d GPU(s), RCCL all-gather of results (BASELINE config 5)" % (args.log2n, world),
                       "n": n, "per_gpu": n // world, "allgather_chunks": chunks, "parallelism": "shard%d" % world},
            "fp6_muls_per_s": fp6_rate,
            "checksum": "%016x" % checksum,
            "identities": n_inf,
            "roofline": {"bound": "valu", "kernel": "k_multiply_ct", "achieved": achieved, "peak": mad_peak,
                         "unit": "Tmad32/s", "frac": achieved / mad_peak, "traffic": traffic,
                         "ms_per_launch": ms_launch, "units_per_launch": units_per_launch, "launches_per_pass": n_launches,
                         "mad32_per_unit": MAD32_PER_CT_MULT,
                         "frac_if_credited_as_ProjectivePoint_multiply": achieved * MAD32_PER_CT_MULT_PROJECTIVE / MAD32_PER_CT_MULT / mad_peak,
                         "note": "achieved = reference-algorithm mad32 count of AffinePoint::multiply (72588 Fp mul x 4) x units / "
                                 "HIP-event launch time; peak = v_mad_u64_u32 issue rate measured live on this device; the kernel "
                                 "is VALU-issue bound (profiles/, DESIGN.md 5): carries and reductions issue at the same rate as the mads; "
                                 "ms_per_launch / units_per_launch are per k_multiply_ct kernel launch (the library bounds each launch); "
                                 "traffic = (FETCH_SIZE + WRITE_SIZE) per unit from the rocprofv3 PMC passes in profiles/ x units_per_launch: "
                                 "the lane's window table is re-read in each of the 64 windows (algorithmic bytes: 225 per unit)"},
            "roofline_fp6_mul": {"bound": "hbm", "kernel": "k_fp6_elementwise<mul>", "achieved": fp6_gbs, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": fp6_gbs / HBM_PEAK_GBS, "traffic": None, "ms_per_launch": fp6_ms,
                                 "units_per_launch": nf, "bytes_per_unit": FP6_MUL_BYTES},
        }
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
            line["cpu_baseline"] = cpu_baseline((1 << args.cpu_sample_log2) * max(1, cores))
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
