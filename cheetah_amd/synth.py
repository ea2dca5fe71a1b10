"""Seeded, counter-based synthetic inputs (host side, numpy).

The same generator is implemented on the device (k_synth_fp6 / k_synth_scalars,
csrc/kernels.hip.h) so that large benchmark inputs never cross PCIe; tests check
the two agree.

    splitmix64(seed, i) = mix(seed + (i + 1) * 0x9E3779B97F4A7C15)

* Fp element   = splitmix64 % p                  (mirrors Fp::random, src/fp.rs:130-132)
* Fp6 element i = draws 6 i .. 6 i + 5           (mirrors Fp6::random, src/fp6.rs:385-394)
* scalar i      = 256-bit draw (limbs 4 i .. 4 i + 3), minus q while >= q, as
                  canonical 32 little-endian bytes (the form Scalar::to_bytes
                  hands to the ladders, src/scalar.rs:477-489)
"""
import numpy as np

P = 0xFFFFFFFF00000001
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
Q_LIMBS = [(Q >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, idx):
    """Vectorised over idx (array of uint64 counters)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def fp6_elements(seed, n, start=0):
    """n canonical Fp6 elements, shape (n, 6) uint64."""
    idx = np.arange(6 * start, 6 * (start + n), dtype=np.uint64)
    return (splitmix64(seed, idx) % np.uint64(P)).reshape(n, 6)


def scalars(seed, n, start=0):
    """n canonical scalars as (n, 32) uint8 little-endian bytes.  Vectorised: 2^256 / q < 3, so
    three conditional subtractions of q (limb-wise, with borrow) reduce any draw."""
    idx = np.arange(4 * start, 4 * (start + n), dtype=np.uint64)
    v = splitmix64(seed, idx).reshape(n, 4).copy()
    ql = [np.uint64(x) for x in Q_LIMBS]
    one, zero = np.uint64(1), np.uint64(0)
    with np.errstate(over="ignore"):
        for _ in range(3):
            ge = np.ones(n, dtype=bool)       # v >= q, decided from the top limb down
            decided = np.zeros(n, dtype=bool)
            for j in (3, 2, 1, 0):
                gt, lt = v[:, j] > ql[j], v[:, j] < ql[j]
                ge = np.where(~decided & lt, False, ge)
                decided |= gt | lt
            borrow = np.zeros(n, dtype=np.uint64)
            for j in range(4):
                sub = ql[j] + borrow              # wraps only when q_j = 2^64 - 1 and borrow = 1 (never: no such limb)
                d = v[:, j] - sub
                nb = np.where(v[:, j] < sub, one, zero)
                v[:, j] = np.where(ge, d, v[:, j])
                borrow = nb
    return np.ascontiguousarray(v).view(np.uint8).reshape(n, 32)


def scalars_reference(seed, n, start=0):
    """The same scalars by Python integers (the definition; slow).  tests/ compare the two."""
    idx = np.arange(4 * start, 4 * (start + n), dtype=np.uint64)
    limbs = splitmix64(seed, idx).reshape(n, 4)
    out = np.empty((n, 4), dtype=np.uint64)
    for i in range(n):
        v = sum(int(limbs[i, j]) << (64 * j) for j in range(4))
        while v >= Q:
            v -= Q
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out.view(np.uint8).reshape(n, 32)
