"""Independent big-integer model of the Cheetah curve arithmetic.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/ and by
tests/golden/generate.py to produce golden vectors.  The product path never
imports it.

This is deliberately NOT a restatement of the reference's code: it is textbook
mathematics on Python integers (schoolbook polynomial multiplication modulo
u^6 - 7, Fermat inversion, the affine chord-and-tangent group law, plain
double-and-add), so that it can pin the C restatement (cheetah_oracle.c) and the
HIP path from an independent direction.  Because affine coordinates are
representation independent, it needs no knowledge of the reference's formulas.
It reproduces the reference's own known-answer vectors (Fp6 mul / square /
inverse, "G is on E", 2G; tests/golden/reference_kats.json) -- see
tests/test_oracle_kat.py.

Constants: p (src/fp.rs:35), u^6 = 7 (src/fp6.rs:40), B = u + 395
(src/curve/mod.rs:42-49), q (src/scalar.rs:41-46), G (src/curve/affine.rs:255-275),
shift point (src/constants.rs:43-61).
"""

P = 0xFFFFFFFF00000001
Q = 0x7AF2599B3B3F22D0563FBF0F990A37B5327AA72330157722D443623EAED4ACCF
COFACTOR = 0x2150B48E071EF610049BC3F5D54304E4A
BETA = 7

ZERO = (0, 0, 0, 0, 0, 0)
ONE = (1, 0, 0, 0, 0, 0)
B = (395, 1, 0, 0, 0, 0)

GENERATOR = (
    (0x263A588F4B0118A1, 0x7757A0BCB26A142D, 0x9215ADFC1E925890,
     0x430AAD2CE14759A4, 0x534ECE54DE4B2C8, 0xB39050F01F7B1F33),
    (0xD57F0D0D47482534, 0x26821D894FA8EA0F, 0xC77F564783EF13A1,
     0x949C360784284EC2, 0xB7040BD639EF3CC4, 0x8AA635F2719D255F),
)
SHIFT_POINT = (
    (0xAB95FB0D188FF2B1, 0xF38263A8C954493E, 0x1BC968D6739028B7,
     0x4F67684FE6B27D1F, 0x6504F8BBBA9077B4, 0x860838116D76E545),
    (0x56988918D5B02CF3, 0xC9FE3EF9D7652C54, 0x4923E90C4D6F4538,
     0xC747AC3F82369F27, 0x6E74C106972C71F5, 0xCF190938F1C64BB5),
)


# ---------------------------------------------------------------- Fp6 ----

def f6(a):
    return tuple(int(x) % P for x in a)


def f6_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f6_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f6_neg(a):
    return tuple((-x) % P for x in a)


def f6_mul(a, b):
    """Schoolbook product of two degree-5 polynomials, reduced by u^6 = 7."""
    t = [0] * 11
    for i in range(6):
        ai = a[i]
        for j in range(6):
            t[i + j] += ai * b[j]
    return tuple((t[k] + (BETA * t[k + 6] if k < 5 else 0)) % P for k in range(6))


def f6_sqr(a):
    return f6_mul(a, a)


def f6_scale(a, k):
    return tuple((x * k) % P for x in a)


def f6_pow(a, e):
    r = ONE
    base = a
    while e:
        if e & 1:
            r = f6_mul(r, base)
        base = f6_mul(base, base)
        e >>= 1
    return r


def f6_inv(a):
    """a^(p^6 - 2); returns None for zero."""
    if a == ZERO:
        return None
    return f6_pow(a, P ** 6 - 2)


def f6_frobenius(a, k=1):
    return f6_pow(a, P ** k)


# -------------------------------------------------- curve, affine model ----
# A point is None (identity) or a pair (x, y) of Fp6 tuples.

def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f6_sqr(y) == f6_add(f6_add(f6_mul(f6_sqr(x), x), x), B)


def aff_neg(pt):
    if pt is None:
        return None
    return (pt[0], f6_neg(pt[1]))


def aff_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 != y2 or y1 == ZERO:
            return None
        num = f6_add(f6_scale(f6_sqr(x1), 3), ONE)  # 3x^2 + a, a = 1
        den = f6_scale(y1, 2)
    else:
        num = f6_sub(y2, y1)
        den = f6_sub(x2, x1)
    lam = f6_mul(num, f6_inv(den))
    x3 = f6_sub(f6_sub(f6_sqr(lam), x1), x2)
    y3 = f6_sub(f6_mul(lam, f6_sub(x1, x3)), y1)
    return (x3, y3)


# Jacobian coordinates only to make scalar multiplication fast in Python
# (one inversion per result).  Textbook dbl-2007-bl / add-2007-bl shapes for a
# general short Weierstrass curve with a = 1.

def _jac_dbl(pt):
    X, Y, Z = pt
    if Z == ZERO or Y == ZERO:
        return (ONE, ONE, ZERO)
    XX = f6_sqr(X)
    YY = f6_sqr(Y)
    YYYY = f6_sqr(YY)
    ZZ = f6_sqr(Z)
    S = f6_scale(f6_mul(X, YY), 4)
    M = f6_add(f6_scale(XX, 3), f6_sqr(ZZ))  # a = 1
    X3 = f6_sub(f6_sqr(M), f6_scale(S, 2))
    Y3 = f6_sub(f6_mul(M, f6_sub(S, X3)), f6_scale(YYYY, 8))
    Z3 = f6_scale(f6_mul(Y, Z), 2)
    return (X3, Y3, Z3)


def _jac_add(p1, p2):
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    if Z1 == ZERO:
        return p2
    if Z2 == ZERO:
        return p1
    Z1Z1 = f6_sqr(Z1)
    Z2Z2 = f6_sqr(Z2)
    U1 = f6_mul(X1, Z2Z2)
    U2 = f6_mul(X2, Z1Z1)
    S1 = f6_mul(f6_mul(Y1, Z2), Z2Z2)
    S2 = f6_mul(f6_mul(Y2, Z1), Z1Z1)
    if U1 == U2:
        if S1 == S2:
            return _jac_dbl(p1)
        return (ONE, ONE, ZERO)
    H = f6_sub(U2, U1)
    R = f6_sub(S2, S1)
    HH = f6_sqr(H)
    HHH = f6_mul(H, HH)
    V = f6_mul(U1, HH)
    X3 = f6_sub(f6_sub(f6_sqr(R), HHH), f6_scale(V, 2))
    Y3 = f6_sub(f6_mul(R, f6_sub(V, X3)), f6_mul(S1, HHH))
    Z3 = f6_mul(f6_mul(Z1, Z2), H)
    return (X3, Y3, Z3)


def _jac_to_aff(pt):
    X, Y, Z = pt
    if Z == ZERO:
        return None
    zi = f6_inv(Z)
    zi2 = f6_sqr(zi)
    return (f6_mul(X, zi2), f6_mul(f6_mul(Y, zi2), zi))


def scalar_mul(k, pt):
    """k * pt for a non-negative integer k (NOT reduced modulo q: the reference
    ladders accept any 32 bytes), affine in / affine out."""
    if pt is None or k == 0:
        return None
    acc = (ONE, ONE, ZERO)
    base = (pt[0], pt[1], ONE)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(acc)
        if bit == "1":
            acc = _jac_add(acc, base)
    return _jac_to_aff(acc)


def scalar_from_bytes(b):
    return int.from_bytes(bytes(b), "little")


# ------------------------------------------------------------ encodings ----

def affine_to_limbs(pt):
    """(x[6] | y[6], infinity) in the ABI's canonical-limb convention; the
    identity is (0, 1, inf = 1) as AffinePoint::identity(), src/curve/affine.rs:224-230."""
    if pt is None:
        return list(ZERO) + list(ONE), 1
    return list(pt[0]) + list(pt[1]), 0


def limbs_to_affine(limbs, inf=0):
    if inf:
        return None
    return (f6(limbs[0:6]), f6(limbs[6:12]))


def proj_to_affine(limbs18):
    x, y, z = f6(limbs18[0:6]), f6(limbs18[6:12]), f6(limbs18[12:18])
    if z == ZERO:
        return None
    zi = f6_inv(z)
    return (f6_mul(x, zi), f6_mul(y, zi))


def jac_to_affine(limbs18):
    return _jac_to_aff((f6(limbs18[0:6]), f6(limbs18[6:12]), f6(limbs18[12:18])))
