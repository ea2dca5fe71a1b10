"""World-size-2 (and 4) rehearsal of the sharding + all-gather layer on CPU with
gloo: the index plan, the block-cyclic layout and the gather must reproduce the
single-rank result on every rank.  The compute step is a stand-in that writes a
function of the GLOBAL index (the real one is the HIP ladder; its per-element
results do not depend on how the batch is sharded)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cheetah_amd.distributed import shard_plan, sharded_apply


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_compute(g0, count, out_xy, out_inf):
    idx = torch.arange(g0, g0 + count, dtype=torch.int64)
    out_xy.copy_(idx[:, None] * 1000 + torch.arange(out_xy.shape[1], dtype=torch.int64)[None, :])
    out_inf.copy_((idx % 7 == 0).to(torch.uint8))


def _expected(n, width):
    xy = torch.empty((n, width), dtype=torch.int64)
    inf = torch.empty((n,), dtype=torch.uint8)
    _fake_compute(0, n, xy, inf)
    return xy, inf


def _worker(rank, world, port, n, chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xy, inf = sharded_apply(_fake_compute, n, 12, rank, world, chunks, "cpu")
        wxy, winf = _expected(n, 12)
        ok = bool(torch.equal(xy, wxy) and torch.equal(inf, winf))
        # no-gather mode leaves only this rank's blocks filled in
        xy2, inf2 = sharded_apply(_fake_compute, n, 12, rank, world, chunks, "cpu", gather=False)
        s, plan = shard_plan(n, world, chunks)
        for c, g0 in plan[rank]:
            ok = ok and bool(torch.equal(xy2[g0:g0 + s], wxy[g0:g0 + s]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 4), (4, 2)])
def test_sharded_apply_gloo(world, chunks):
    n = 64 * world * chunks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True) for r in range(world)]


def test_shard_plan_covers_every_index_once():
    for n, world, chunks in [(64, 1, 1), (1024, 8, 8), (96, 2, 3)]:
        s, plan = shard_plan(n, world, chunks)
        seen = sorted(g for r in range(world) for (c, g0) in plan[r] for g in range(g0, g0 + s))
        assert seen == list(range(n))
        # the all-gather of chunk c fills one contiguous slab, rank-major
        for r in range(world):
            for c, g0 in plan[r]:
                assert g0 == c * (n // chunks) + r * s
    with pytest.raises(ValueError):
        shard_plan(100, 8, 8)


def test_single_rank_needs_no_process_group():
    xy, inf = sharded_apply(_fake_compute, 128, 12, 0, 1, 1, "cpu")
    wxy, winf = _expected(128, 12)
    assert torch.equal(xy, wxy) and torch.equal(inf, winf)
