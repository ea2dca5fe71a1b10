#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): variable-base constant-time Cheetah scalar
multiplications per second, plus Fp6 multiplications per second.

    python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment this process is only a
launcher: before anything touches the GPU it starts the N rank processes as
fresh children (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...` of this same file), forwards rank 0's JSON line and
exits with the children's status.  Started by torch.distributed.run itself
(WORLD_SIZE set) it is one of the ranks.

Workload (BASELINE config 5): 2^26 seeded (P_i, k_i) pairs, out[i] = k_i * P_i by
AffinePoint::multiply (src/curve/affine.rs:373-375 -> JacobianPoint::multiply,
src/curve/jacobian.rs:599-611), affine results.  The 2^26 pairs are sharded over
the N ranks (strong scaling: the total is fixed) and the affine results are
all-gathered over RCCL so that every rank holds all of them.  Inputs are
synthetic, generated on the device (points r_i * G, seeded scalars) before the
timed region; a step is one full pass.  One JSON line on stdout from rank 0.

--rehearse-cpu runs the SAME launcher, sharding, all-gather and JSON plumbing on
CPU over gloo with a stand-in for the arithmetic (a function of the global
index): a rehearsal of the multi-process path for boxes without GPUs
(tests/test_bench_launcher_cpu.py).  Its line says so and is not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# reference-algorithm work per unit (SURVEY.md 8d / BASELINE.md 2; derivation in DESIGN.md 5).
# Affine in / affine out is AffinePoint::multiply, which runs JacobianPoint::multiply
# (src/curve/affine.rs:373-375, src/curve/jacobian.rs:599-611):
#   LookupTable<16>::from(&JacobianPoint): normalise 300 + 15 complete adds x 300 + batch_normalize(16) 2 103 = 6 903
#   64 windows x (4 x 180 + 297) = 65 088;  - 2^256 S correction 297;  final normalisation 300
FPMUL_PER_CT_MULT = 72588
MAD32_PER_CT_MULT = 4 * FPMUL_PER_CT_MULT   # 1 Fp mul = 4 v_mad_u64_u32 on 32-bit limbs
MAD32_PER_CT_MULT_PROJECTIVE = 348468       # ProjectivePoint::multiply, for comparison only
FP6_MUL_BYTES = 144             # 2 x 48 in + 48 out
BN_JAC_BYTES = 241              # 144 in + 96 + 1 out
BN_JAC_MAD32 = 660              # 165 Fp mul per point (6 M + 1 S), the chunk's inversion not credited
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md


def ceiling_model():
    """Floor of THIS formulation per scalar multiplication, in v_mad_u64_u32 issue slots: what the
    kernel would take if every Fp6 operation ran at its instruction floor, nothing else were executed and the SIMDs
    never waited.  Instruction classes are priced by their measured issue time relative to the mad
    (profiles/r01_valu_mix_w4.json: a mad alone 2.11 ns per wave-instruction; alternating with carry-class
    instructions, as everywhere in this arithmetic, the pair takes 3.70 ns, i.e. the carry-class instruction 1.59 ns).
      Fp6 mul    = 36 products x (4 mad + 4 carry counts) + 6 reductions x (1 mad + 13) + 5 scalings by 7 x 9
                 = 150 mad + 267 carry-class  -> 351 mad slots   (credited: 96)
      Fp6 square = 21 x (4 + 4) + 6 x (1 + 13) + 5 x 9 + 5 doublings x 6 = 90 mad + 237 carry-class -> 269 (credited: 84)
      Fp6 add / sub / double = 36 carry-class -> 27
    Per window: 4 doublings (4 M + 4 S, the last without its z4 product; 8 linear operations each) + 1 mixed addition
    (8 M + 5 S, 8 linear operations, the constant-time scan's 192 selects); 64 windows; the 8-entry table (26 M + 11 S
    + 3 Fp6 inversions of ~3 600 slots each) and the final normalisation (an inversion + 4 M + S) ~ 36 k slots.
    ceiling_frac = credited mad32 / those slots."""
    half = (3.70 - 2.11) / 2.11
    M = 150 + 267 * half
    S = 90 + 237 * half
    L = 36 * half
    dbl = 4 * M + 4 * S + 8 * L
    add = 8 * M + 5 * S + 8 * L + 192 * half
    window = 4 * dbl - M + add
    slots = 64 * window + 36000
    return {"min_mad_slots_per_unit": round(slots), "ceiling_frac": MAD32_PER_CT_MULT / slots}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=26, help="total scalar multiplications = 2^log2n (26 = BASELINE config 5)")
    ap.add_argument("--chunks", type=int, default=0, help="all-gather chunks per rank (0: 1 at N=1, 8 otherwise)")
    ap.add_argument("--cpu-sample-log2", type=int, default=14, help="CPU baseline sample = 2^k multiplications per core")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Fp6 / batch_normalize roofline measurements")
    ap.add_argument("--launch-rounds", type=int, default=0,
                    help="rounds of the resident grid per ladder launch (0: 16 on one GPU, where no collective waits for a "
                         "launch boundary; the library's default 2 otherwise)")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="CUs' worth of workgroup slots the persistent launches leave free for the RCCL kernel (--gpus > 1)")
    ap.add_argument("--rccl-at-one", action="store_true",
                    help="--gpus 1 only: run the multi-GPU code path (NCCL process group with its options, chunked in-place "
                         "all-gathers from the side stream, all-reduces) in a one-rank group -- execution evidence for that "
                         "path on a one-GPU box, not a scaling measurement")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="gloo on CPU with a stand-in for the arithmetic: launcher / sharding / gather rehearsal, not a measurement")
    return ap.parse_args()


# ------------------------------------------------------------------ launcher -----

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """Parent of a multi-GPU run: starts the rank processes and forwards rank 0's line.  Nothing here
    initialises the GPU (torch.cuda.device_count() does not), so the children are the first to."""
    if not args.rehearse_cpu:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, ndev))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if p.returncode != 0:
        sys.stderr.write("bench.py: a rank failed (torch.distributed.run exit %d)\n" % p.returncode)
        return p.returncode or 1
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        return 1
    print(lines[0], flush=True)
    return 0


# --------------------------------------------------------------- CPU baseline -----

def cpu_baseline(n_sample, seed=5):
    """The oracle (C restatement of the reference CPU path, OpenMP over independent
    elements) timed on this box's host cores on a bounded sample of the same workload."""
    import numpy as np
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


# ------------------------------------------------------------------- one rank -----

def rehearse(args, world, rank):
    """gloo / CPU rehearsal: same plan, same sharded_apply, same line; the 'arithmetic' is a stand-in."""
    import torch
    import torch.distributed as dist
    from cheetah_amd.distributed import sharded_apply
    if world > 1:
        dist.init_process_group("gloo")
    n = 1 << args.log2n
    chunks = args.chunks or (1 if world == 1 else 8)

    def compute(g0, cnt, oxy, oinf):
        idx = torch.arange(g0, g0 + cnt, dtype=torch.int64)
        oxy.copy_(idx[:, None] * 12 + torch.arange(12, dtype=torch.int64)[None, :])
        oinf.copy_((idx % 5 == 0).to(torch.uint8))

    for _ in range(args.warmup):
        sharded_apply(compute, n, 12, rank, world, chunks, "cpu")
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        xy, inf = sharded_apply(compute, n, 12, rank, world, chunks, "cpu")
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ones = torch.ones(1, dtype=torch.int64)
    # the same pass without the gather (report both)
    t1 = time.perf_counter()
    sharded_apply(compute, n, 12, rank, world, chunks, "cpu", gather=False)
    if world > 1:
        dist.barrier()
    dt_ng = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt, dt_ng], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_ng = float(t[0].item()), float(t[1].item())
        dist.all_reduce(ones)
    want = torch.arange(n * 12, dtype=torch.int64).reshape(n, 12)
    ok = bool(torch.equal(xy, want)) and int(inf.sum()) == len(range(0, n, 5))
    if rank == 0:
        print(json.dumps({
            "metric": "Cheetah variable-base scalar mults/sec (1 GPU, node) + Fp6 muls/sec", "value": n * args.steps / dt,
            "unit": "scalar_mults/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "REHEARSAL on CPU over gloo: stand-in for the arithmetic, not a measurement",
            "rehearsal": True, "ranks_seen": int(ones.item()), "gathered_ok": ok,
            "compute_ms": dt_ng * 1e3, "no_gather_value": n / dt_ng, "gather_ms_exposed": dt / args.steps * 1e3 - dt_ng * 1e3,
            "checksum": "%016x" % (int(xy.sum().item()) & 0xFFFFFFFFFFFFFFFF),
            "config": {"workload": "launcher / sharding / all-gather rehearsal", "n": n, "per_gpu": n // world,
                       "allgather_chunks": chunks, "parallelism": "shard%d" % world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def secondary_rooflines(eng, torch):
    """Config 1 (elementwise Fp6 multiplication) and config 4 (batch_normalize, 2^24 Jacobian points):
    HIP events on the launch stream, one GPU."""
    from cheetah_amd import AFFINE, JACOBIAN
    out = {}

    def timed(f, reps):
        f()
        torch.cuda.synchronize()
        eng._use_torch_stream()
        eng.timer_start()
        for _ in range(reps):
            f()
        return eng.timer_stop_ms() / reps

    def timed_each(f, reps):
        """(mean, min) over separately timed repetitions: these kernels are a few ms, and this part of the run follows
        44 s of ladder work, so the clocks the device sustains move the figure by several per cent from call to call"""
        f()
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            eng._use_torch_stream()
            eng.timer_start()
            f()
            ts.append(eng.timer_stop_ms())
        return sum(ts) / len(ts), min(ts)

    # config 1 as BASELINE.json sizes it: 2^20 pairs = 151 MB, which stays in the 256 MB Infinity Cache between
    # repetitions -- NOT an HBM figure
    nf = 1 << 20
    fa, fb = eng.synth_fp6(1, nf, 0), eng.synth_fp6(1, nf, nf)
    ms20 = timed(lambda: eng.fp6_mul_batch(fa, fb), 20)
    out["fp6_muls_per_s"] = nf / (ms20 * 1e-3)
    out["fp6_mul_2p20_infinity_cache_resident"] = {"ms_per_launch": ms20, "GBps": nf * FP6_MUL_BYTES / (ms20 * 1e-3) / 1e9,
                                                   "note": "151 MB working set re-run back to back: served by the 256 MB Infinity Cache"}
    del fa, fb
    # the HBM-bound figure: 2^24 pairs = 2.4 GB per launch, nothing survives in a cache from one launch to the next
    nc = 1 << 24
    ca, cb = eng.synth_fp6(1, nc, 0), eng.synth_fp6(1, nc, nc)
    msc = timed(lambda: eng.fp6_mul_batch(ca, cb), 5)
    gbs = nc * FP6_MUL_BYTES / (msc * 1e-3) / 1e9
    out["fp6_muls_per_s_cold_2p24"] = nc / (msc * 1e-3)
    out["roofline_fp6_mul"] = {"bound": "hbm", "kernel": "k_fp6_elementwise<mul>", "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "ms_per_launch": msc,
                               "units_per_launch": nc, "bytes_per_unit": FP6_MUL_BYTES,
                               "note": "cold: 2^24 pairs (2.4 GB per launch, 9x the Infinity Cache)"}
    del ca, cb
    # config 4: 2^24 Jacobian points (x l^2, y l^3, l) from k_i G, an identity every 1000
    nb = 1 << 24
    aff, _ = eng.mul_by_generator_batch(eng.synth_scalars(4, nb), AFFINE)
    lam = eng.synth_fp6(4, nb)
    l2 = eng.fp6_square_batch(lam)
    l3 = eng.fp6_mul_batch(l2, lam)
    affi = aff.view(torch.int64)
    x = eng.fp6_mul_batch(affi[:, 0:6].contiguous(), l2)
    y = eng.fp6_mul_batch(affi[:, 6:12].contiguous(), l3)
    jac = torch.cat([x.view(torch.int64), y.view(torch.int64), lam.view(torch.int64)], dim=1).contiguous()
    del x, y, l2, l3, lam
    ms_free, ms_free_min = timed_each(lambda: eng.batch_normalize(jac, JACOBIAN), 8)   # the same batch before the identities go in
    jac[torch.arange(0, nb, 1000, device=jac.device), 12:18] = 0
    msb, msb_min = timed_each(lambda: eng.batch_normalize(jac, JACOBIAN), 8)
    xy, inf = eng.batch_normalize(jac, JACOBIAN)
    want = affi.clone()
    want[torch.arange(0, nb, 1000, device=jac.device)] = torch.tensor([0] * 6 + [1] + [0] * 5, dtype=torch.int64, device=jac.device)
    bn_ok = bool(torch.equal(xy.view(torch.int64), want)) and int(inf.sum()) == len(range(0, nb, 1000))
    mad_peak = eng.calibrate_mad_peak()
    ach = nb * BN_JAC_MAD32 / (msb * 1e-3) / 1e12
    out["batch_normalize_points_per_s"] = nb / (msb * 1e-3)
    out["roofline_batch_normalize"] = {"bound": "valu", "kernel": "k_batch_normalize<jacobian>", "achieved": ach, "peak": mad_peak,
                                       "unit": "Tmad32/s", "frac": ach / mad_peak, "traffic": None, "ms_per_launch": msb,
                                       "units_per_launch": nb, "mad32_per_unit": BN_JAC_MAD32,
                                       "hbm_GBps": nb * BN_JAC_BYTES / (msb * 1e-3) / 1e9, "hbm_frac": nb * BN_JAC_BYTES / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "round_trip_ok": bn_ok,
                                       "best_ms_per_launch": msb_min, "best_points_per_s": nb / (msb_min * 1e-3),
                                       "identity_free_batch": {"ms_per_launch": ms_free, "points_per_s": nb / (ms_free * 1e-3),
                                                               "best_ms_per_launch": ms_free_min},
                                       "note": "config 4: 2^24 Jacobian points, identities included; output compared in full with the affine points they were built from"}
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rehearse_cpu:
        sys.exit(rehearse(args, world, rank))

    import torch
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: no GPU %d (visible: %d)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    import torch.distributed as dist
    dist_on = world > 1 or args.rccl_at_one
    if dist_on:
        kw = {}
        if world == 1:  # --rccl-at-one: no launcher set the rendezvous up
            kw = {"init_method": "tcp://127.0.0.1:%d" % _free_port(), "rank": 0, "world_size": 1}
        # the collective's kernel runs on ProcessGroupNCCL's own stream: ask for that one to be high priority
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            dist.init_process_group("nccl", device_id=torch.device(device), pg_options=opts, **kw)
        except (AttributeError, TypeError, ValueError):
            dist.init_process_group("nccl", device_id=torch.device(device), **kw)

    from cheetah_amd import AFFINE, Engine
    from cheetah_amd.distributed import shard_plan, sharded_apply

    eng = Engine(local_rank)
    if dist_on and args.reserve_cus:
        eng.set_reserved_cus(args.reserve_cus)
    launch_rounds = args.launch_rounds or (2 if dist_on else 16)
    eng.set_launch_rounds(launch_rounds)
    n = 1 << args.log2n
    chunks = args.chunks or (8 if dist_on else 1)
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

    def compute(g0, cnt, oxy, oinf):
        j = local_of[g0]
        eng.multiply_batch(pts[j * s:(j + 1) * s], ks[j * s:(j + 1) * s], AFFINE, AFFINE, out=oxy, inf=oinf)

    def step(gather=True):
        return sharded_apply(compute, n, 12, rank, world, chunks, device, gather=gather, collective_at_one=args.rccl_at_one)

    def timed_kernel_pass():
        # one un-overlapped pass, HIP events on the launch stream around each call's kernel launches
        total, launches = 0.0, 0
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

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not dist_on:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        out = step()
        del out
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    xy, inf = out
    checksum = int(xy.sum().item()) & 0xFFFFFFFFFFFFFFFF  # wrap-around sum of canonical limbs, identical on every N
    n_inf = int(inf.sum().item())
    del out, xy, inf

    # ---- what the gather costs: the same pass without it (this rank's blocks only) --------
    ones = torch.ones(1, dtype=torch.int64, device=device)
    if dist_on:
        dist.all_reduce(ones)                     # ranks_seen: a sum of ones over RCCL
    barrier()
    t0 = time.perf_counter()
    o = step(gather=False)                        # every N, N = 1 included (report both)
    barrier()
    compute_ms = max_over_ranks(time.perf_counter() - t0) * 1e3
    del o

    # ---- roofline of the dominant kernel (k_multiply_ct), measured live ---------
    ms_launch, units_per_launch, n_launches = timed_kernel_pass()   # averages over the pass's kernel launches
    mad_peak = eng.calibrate_mad_peak()          # T v_mad_u64_u32 / s, measured on this device
    achieved = units_per_launch * MAD32_PER_CT_MULT / (ms_launch * 1e-3) / 1e12
    traffic = None                               # HBM bytes per launch from the committed PMC passes (profiles/)
    for name in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tr = json.load(f)["k_multiply_ct"]
            traffic = (tr["fetch_bytes_per_unit"] + tr["write_bytes_per_unit"]) * units_per_launch
            break
        except (OSError, KeyError, ValueError):
            pass

    secondary = {}
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = secondary_rooflines(eng, torch)

    if rank == 0:
        total = n * args.steps
        ceil = ceiling_model()
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
                       "n": n, "per_gpu": n // world, "allgather_chunks": chunks, "parallelism": "shard%d" % world,
                       "reserved_cus": args.reserve_cus if dist_on else 0, "launch_rounds": launch_rounds},
            "checksum": "%016x" % checksum,
            "identities": n_inf,
            "ranks_seen": int(ones.item()),
            "roofline": {"bound": "valu", "kernel": "k_multiply_ct", "achieved": achieved, "peak": mad_peak,
                         "unit": "Tmad32/s", "frac": achieved / mad_peak, "traffic": traffic,
                         "ms_per_launch": ms_launch, "units_per_launch": units_per_launch, "launches_per_pass": n_launches,
                         "mad32_per_unit": MAD32_PER_CT_MULT,
                         "ceiling_frac": ceil["ceiling_frac"], "min_mad_slots_per_unit": ceil["min_mad_slots_per_unit"],
                         "frac_if_credited_as_ProjectivePoint_multiply": achieved * MAD32_PER_CT_MULT_PROJECTIVE / MAD32_PER_CT_MULT / mad_peak,
                         "note": "achieved = reference-algorithm mad32 count of AffinePoint::multiply (72588 Fp mul x 4) x units / "
                                 "HIP-event launch time; peak = v_mad_u64_u32 issue rate measured live on this device; the kernel "
                                 "is VALU-issue bound (profiles/, DESIGN.md 2, 5): a carry-class instruction takes 0.85-0.95 of a mad's issue time and there are as many of them; "
                                 "ceiling_frac = credited mad32 / the formulation's floor in mad issue slots (bench.py ceiling_model, DESIGN.md 5); "
                                 "ms_per_launch / units_per_launch are per k_multiply_ct kernel launch (the library bounds each launch); "
                                 "traffic = (FETCH_SIZE + WRITE_SIZE) per unit from the rocprofv3 PMC passes in profiles/ x units_per_launch: "
                                 "the lane's window table is re-read in each of the 64 windows (algorithmic bytes: 225 per unit)"},
        }
        if args.rccl_at_one:
            line["rccl_at_one"] = True
        # the no-gather mode (results stay sharded on the ranks that computed them): one pass, max over ranks
        line["compute_ms"] = compute_ms
        line["no_gather_value"] = n / (compute_ms * 1e-3)
        line["gather_ms_exposed"] = dt / args.steps * 1e3 - compute_ms
        line.update(secondary)
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
            line["cpu_baseline"] = cpu_baseline((1 << args.cpu_sample_log2) * max(1, cores))
        print(json.dumps(line), flush=True)
    eng.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
