"""Sharding of independent scalar multiplications over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
reference has nothing distributed; the only exchange step the path has is
returning results, so there is exactly one kind of collective here: an
all-gather of the affine results over xGMI.  There is no exchange during
compute.

Index layout (block-cyclic so that every all-gather writes one contiguous slab
of the result and needs no staging copy):

    global index g in [0, n):   chunk c = g // (n / C)
                                rank  r = (g % (n / C)) // s,   s = n / (C * W)
    rank r computes, for every chunk c, the s consecutive elements starting at
    c * (n / C) + r * s; the all-gather of chunk c fills [c * n / C, (c + 1) * n / C).

Chunk c's all-gather is enqueued from a second stream while chunk c + 1
computes; only the last chunk's gather (1 / C of the traffic) is exposed.
ProcessGroupNCCL runs the collective's kernel on a stream of its own that waits
for the enqueuing stream; whether that stream is high priority is a property of
the process group (bench.py asks for it with is_high_priority_stream), not of
the stream used here.  The ladder kernels fill every CU with persistent
workgroups, so the library bounds each launch to two rounds of the resident grid
(18 ms measured at 2^19 multiplications): a collective's workgroups get CU slots
at the next launch boundary at the latest, and nothing computes after the last
chunk, so the exposed gather is the last chunk's own duration.
"""
import torch
import torch.distributed as dist


def shard_plan(n, world, chunks):
    """Returns (s, [(chunk, global_start) ...] for each rank) -- pure index logic."""
    if n % (world * chunks) != 0:
        raise ValueError("n = %d must be divisible by world * chunks = %d" % (n, world * chunks))
    s = n // (world * chunks)
    per_chunk = n // chunks
    return s, [[(c, c * per_chunk + r * s) for c in range(chunks)] for r in range(world)]


def sharded_apply(compute, n, width, rank, world, chunks, device, dtype=torch.int64, group=None, gather=True,
                  collective_at_one=False):
    """Runs compute(global_start, count, out_xy_view, out_inf_view) for this rank's
    blocks and all-gathers the results.  Returns (xy, inf) holding all n results
    on every rank (gather=True), or with only this rank's blocks filled in.

    compute must enqueue its work on the CURRENT stream of `device`.
    collective_at_one: issue the collectives even in a one-rank group (bench.py --rccl-at-one: the
    RCCL code path on a one-GPU box).
    """
    s, plan = shard_plan(n, world, chunks)
    per_chunk = n // chunks
    xy = torch.empty((n, width), dtype=dtype, device=device)
    inf = torch.empty((n,), dtype=torch.uint8, device=device)
    use_cuda = torch.device(device).type == "cuda"
    collective = gather and (world > 1 or collective_at_one)
    # a side stream orders "gather chunk c after its compute" without holding back chunk c + 1's launches
    comm = torch.cuda.Stream(device=device) if (use_cuda and collective) else None
    main = torch.cuda.current_stream(device) if use_cuda else None
    for c, g0 in plan[rank]:
        compute(g0, s, xy[g0:g0 + s], inf[g0:g0 + s])
        if collective:
            lo = c * per_chunk
            if comm is not None:
                comm.wait_stream(main)
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(xy[lo:lo + per_chunk], xy[g0:g0 + s], group=group)
                    dist.all_gather_into_tensor(inf[lo:lo + per_chunk], inf[g0:g0 + s], group=group)
            else:  # CPU / gloo rehearsal of the same index logic
                outs = [xy[lo + r * s: lo + (r + 1) * s] for r in range(world)]
                dist.all_gather(outs, xy[g0:g0 + s].clone(), group=group)
                outs = [inf[lo + r * s: lo + (r + 1) * s] for r in range(world)]
                dist.all_gather(outs, inf[g0:g0 + s].clone(), group=group)
    if comm is not None:
        main.wait_stream(comm)
    return xy, inf
