#!/usr/bin/env python3
"""Full-size golden sums for the BASELINE configurations, produced by the ORACLE on the CPU.

    python tests/golden/generate_fullsize.py config5            # ~25 min on 8 host threads
    python tests/golden/generate_fullsize.py config2 config3 config4

For each configuration the oracle (oracle/cheetah_oracle.c, the C restatement of the reference's
CPU path) is run over the WHOLE seeded input of that configuration, block by block, and the
wrap-around 64-bit sum of all canonical output limbs of every 2^20-element block is written to
tests/golden/fullsize_<config>.json, plus the sum of sums.  The GPU tests recompute the same block
sums from the HIP path's output (tests/test_gpu_parity.py): a block sum that differs names the block.
Nothing here touches the GPU or cheetah_amd's product path; cheetah_amd.synth is the seeded input
generator (host side).

Inputs (the same streams bench.py and the GPU tests use):
  config2  k_i = scalars(seed 2, i),                i < 2^22: out = k_i G                (BasePointTable::multiply, affine)
  config3  P_i = config2's outputs, k_i = scalars(3, i), i < 2^22: out = k_i P_i         (multiply_vartime, affine)
  config4  a_i = scalars(4, i) G, l_i = fp6(4, i),   i < 2^24: Jacobian (x l^2, y l^3, l), z = 0 at every 1000th:
           out = batch_normalize                                                          (JacobianPoint::batch_normalize)
  config5  P_i = scalars(5, 2^26 + i) G, k_i = scalars(5, i), i < 2^26: out = k_i P_i    (AffinePoint::multiply)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cheetah_amd import synth  # noqa: E402
from oracle import binding as ora  # noqa: E402

BLOCK = 1 << 20
MASK = 0xFFFFFFFFFFFFFFFF


def wsum(a):
    """wrap-around 64-bit sum of all limbs"""
    with np.errstate(over="ignore"):
        return int(np.asarray(a, dtype=np.uint64).sum(dtype=np.uint64))


def emit(name, what, sums, infs, extra=None):
    total = 0
    for s in sums:
        total = (total + s) & MASK
    doc = {"config": name, "what": what, "block": BLOCK, "blocks": len(sums),
           "block_sums": ["%016x" % s for s in sums], "identities_per_block": infs, "checksum": "%016x" % total,
           "pinned_by": "oracle/cheetah_oracle.c run over the full input by tests/golden/generate_fullsize.py (CPU, no GPU involved)"}
    if extra:
        doc.update(extra)
    path = os.path.join(HERE, "fullsize_%s.json" % name)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote", path, "checksum", doc["checksum"], flush=True)


def config5(log2n=26):
    n = 1 << log2n
    sums, infs = [], []
    t0 = time.time()
    part = os.path.join(HERE, "fullsize_config5_2p%d.partial.json" % log2n)   # resumable: the run is ~10 core-minutes per block
    if os.path.exists(part):
        with open(part) as f:
            d = json.load(f)
        sums, infs = d["sums"], d["infs"]
    for b in range(len(sums), n // BLOCK):
        g0 = b * BLOCK
        r = synth.scalars(5, BLOCK, (1 << 26) + g0)
        pts, pinf = ora.mul_by_generator_batch(r, 0)
        assert not pinf.any()
        k = synth.scalars(5, BLOCK, g0)
        out, inf = ora.multiply_batch(pts, 0, k, 0)
        sums.append(wsum(out))
        infs.append(int(inf.sum()))
        with open(part, "w") as f:
            json.dump({"sums": sums, "infs": infs}, f)
        print("config5 block %d/%d  %.0f s" % (b + 1, n // BLOCK, time.time() - t0), flush=True)
    os.remove(part)
    emit("config5" if log2n == 26 else "config5_2p%d" % log2n,
         "wrap-around 64-bit sum of the canonical limbs of out[i] = k_i P_i per 2^20-block, P_i = scalars(5, 2^26 + i) G, "
         "k_i = scalars(5, i), i < 2^%d (AffinePoint::multiply, src/curve/affine.rs:373-375)" % log2n, sums, infs)


def config2_3():
    n = 1 << 22
    s2, i2, s3, i3 = [], [], [], []
    for b in range(n // BLOCK):
        g0 = b * BLOCK
        k2 = synth.scalars(2, BLOCK, g0)
        pts, pinf = ora.mul_by_generator_batch(k2, 0)
        s2.append(wsum(pts))
        i2.append(int(pinf.sum()))
        k3 = synth.scalars(3, BLOCK, g0)
        out, inf = ora.multiply_batch(pts, 0, k3, 0, vartime=True)
        s3.append(wsum(out))
        i3.append(int(inf.sum()))
        print("config2/3 block %d/%d" % (b + 1, n // BLOCK), flush=True)
    emit("config2", "per-2^20-block sums of the canonical limbs of k_i G, k_i = scalars(2, i), i < 2^22 "
         "(BasePointTable::multiply, src/lookup.rs:189-206, affine)", s2, i2)
    emit("config3", "per-2^20-block sums of the canonical limbs of k_i P_i, P_i = config 2's outputs, k_i = scalars(3, i), "
         "i < 2^22 (AffinePoint::multiply_vartime, src/curve/affine.rs:384-386)", s3, i3)


def config4():
    n = 1 << 24
    sums, infs = [], []
    for b in range(n // BLOCK):
        g0 = b * BLOCK
        aff, ainf = ora.mul_by_generator_batch(synth.scalars(4, BLOCK, g0), 0)
        assert not ainf.any()
        lam = synth.fp6_elements(4, BLOCK, g0)
        l2 = ora.fp6_square_batch(lam)
        l3 = ora.fp6_mul_batch(l2, lam)
        x = ora.fp6_mul_batch(np.ascontiguousarray(aff[:, 0:6]), l2)
        y = ora.fp6_mul_batch(np.ascontiguousarray(aff[:, 6:12]), l3)
        jac = np.concatenate([ora.canon(x), ora.canon(y), lam], axis=1)
        first = (-g0) % 1000
        jac[first::1000, 12:18] = 0
        xy, inf = ora.batch_normalize(jac, True)
        sums.append(wsum(xy))
        infs.append(int(inf.sum()))
        print("config4 block %d/%d" % (b + 1, n // BLOCK), flush=True)
    emit("config4", "per-2^20-block sums of the canonical limbs of JacobianPoint::batch_normalize (src/curve/jacobian.rs:886-920) over "
         "(x l^2, y l^3, l), (x, y) = scalars(4, i) G, l = fp6(4, i), z = 0 at every 1000th index, i < 2^24", sums, infs)


if __name__ == "__main__":
    ora.build()
    todo = sys.argv[1:] or ["config2", "config3", "config4", "config5"]
    if "config2" in todo or "config3" in todo:
        config2_3()
    if "config4" in todo:
        config4()
    for t in todo:
        if t.startswith("config5"):
            config5(int(t.split(":")[1]) if ":" in t else 26)
