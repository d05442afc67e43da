"""CPU check of the *product's* field/curve templates (csrc/ff.cuh, csrc/ec.cuh) through their
__host__ instantiation (tests/csrc/ff_host_shim.hip): the same source the gfx950 kernels compile,
checked against the golden vectors.  No GPU needed."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import pyref as P
from helpers import *

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "csrc", "ff_host_shim.hip")
OUT = os.path.join(ROOT, "tests", "csrc", "build", "libff_host_shim.so")
INC = os.path.join(ROOT, "zksnark-finalproject_amd", "csrc")


@pytest.fixture(scope="module")
def shim():
    deps = [SRC, os.path.join(INC, "ff.cuh"), os.path.join(INC, "ec.cuh")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["hipcc", "--offload-host-only", "-O2", "-shared", "-fPIC", "-I", INC, "-o", OUT, SRC])
    return C.CDLL(OUT)


def u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).view(np.uint32)


def call_field(shim, name, op, a, b):
    a32, b32 = u32(a), u32(b)
    o = np.zeros_like(a32)
    getattr(shim, name)(op, a32.ctypes.data_as(C.c_void_p), b32.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    return o.view(np.uint64)


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_field_kat_host(shim, field):
    nl, to_m, from_m = (4, P.fr_to_mont, P.fr_from_mont) if field == "fr" else (6, P.fq_to_mont, P.fq_from_mont)
    for v in load("field_kat.json")[field]:
        a, b = limbs(to_m(H(v["a"])), nl), limbs(to_m(H(v["b"])), nl)
        for op, key in ((0, "add"), (1, "sub"), (2, "mul")):
            assert from_m(unlimbs(call_field(shim, "ht_%s_op" % field, op, a, b))) == H(v[key]), (field, key, v)
        assert from_m(unlimbs(call_field(shim, "ht_%s_op" % field, 4, a, a))) == H(v["inv_a"])
        assert from_m(unlimbs(call_field(shim, "ht_%s_op" % field, 3, a, a))) == H(v["a"]) ** 2 % (P.R_MOD if field == "fr" else P.Q_MOD)
        assert from_m(unlimbs(call_field(shim, "ht_%s_op" % field, 5, a, a))) == (-H(v["a"])) % (P.R_MOD if field == "fr" else P.Q_MOD)


def test_field_random_vs_python(shim):
    rng = random.Random(11)
    for field, mod, nl, to_m, from_m in (("fr", P.R_MOD, 4, P.fr_to_mont, P.fr_from_mont), ("fq", P.Q_MOD, 6, P.fq_to_mont, P.fq_from_mont)):
        for _ in range(300):
            a, b = rng.randrange(mod), rng.randrange(mod)
            got = from_m(unlimbs(call_field(shim, "ht_%s_op" % field, 2, limbs(to_m(a), nl), limbs(to_m(b), nl))))
            assert got == a * b % mod
        # values just below the modulus / with all-ones limbs exercise the carry paths
        for a in (mod - 1, mod - 2, (1 << (32 * 2 * nl - 1)) % mod, ((1 << 64 * nl) - 1) % mod):
            for b in (mod - 1, 1, 2, a):
                got = from_m(unlimbs(call_field(shim, "ht_%s_op" % field, 2, limbs(to_m(a), nl), limbs(to_m(b), nl))))
                assert got == a * b % mod


def test_fq2_kat_host(shim):
    enc = lambda p: np.concatenate([fq_mont(H(p[0])), fq_mont(H(p[1]))])
    dec = lambda a: [P.fq_from_mont(unlimbs(a[:6])), P.fq_from_mont(unlimbs(a[6:]))]
    for v in load("field_kat.json")["fq2"]:
        a, b = enc(v["a"]), enc(v["b"])
        assert dec(call_field(shim, "ht_fq2_op", 2, a, b)) == [H(x) for x in v["mul"]]
        assert dec(call_field(shim, "ht_fq2_op", 3, a, a)) == [H(x) for x in v["sqr"]]
        assert dec(call_field(shim, "ht_fq2_op", 4, a, a)) == [H(x) for x in v["inv_a"]]


def point_op(shim, group, op, acc, q=None, k=None, neg=0):
    w = 12 if group == "g1" else 24
    acc32 = u32(acc)
    q32 = u32(q if q is not None else np.zeros(4 * w // 1, dtype=np.uint64))
    k32 = u32(k if k is not None else np.zeros(4, dtype=np.uint64))
    ox = np.zeros(4 * w, dtype=np.uint32)
    oa = np.zeros(2 * w, dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    getattr(shim, "ht_%s_op" % group)(op, p(acc32), p(q32), p(k32), neg, p(ox), p(oa))
    return ox.view(np.uint64), oa.view(np.uint64)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_curve_kat_host(shim, group):
    kat = load("curve_kat.json")
    w = 12 if group == "g1" else 24
    gen = G1_GEN_LIMBS if group == "g1" else G2_GEN_LIMBS
    enc = g1_limbs if group == "g1" else g2_limbs
    inf_x = np.zeros(2 * w, dtype=np.uint64)           # XYZZ infinity (zz = 0)
    gen_x, _ = point_op(shim, group, 4, inf_x, q=gen)   # from_affine
    for v in kat[group + "_mul"]:
        _, aff = point_op(shim, group, 3, gen_x, k=fr_canon(H(v["k"])))
        exp, einf = enc(v["p"])
        assert np.array_equal(aff, exp), v["k"]          # infinity encodes as (0,0) == zeros
    for v in kat[group + "_add"]:
        ax, aa = point_op(shim, group, 3, gen_x, k=fr_canon(H(v["a"])))
        bx, ba = point_op(shim, group, 3, gen_x, k=fr_canon(H(v["b"])))
        exp, _ = enc(v["p"])
        _, got_mixed = point_op(shim, group, 0, ax, q=ba)          # xyzz += affine
        _, got_full = point_op(shim, group, 1, ax, q=bx)           # xyzz += xyzz
        assert np.array_equal(got_mixed, exp), v
        assert np.array_equal(got_full, exp), v
        # negated mixed add: a + (-(−b)) style check: a - b + b
        cx, _ = point_op(shim, group, 0, ax, q=ba, neg=1)
        _, back = point_op(shim, group, 0, cx, q=ba)
        assert np.array_equal(back, aa)


# ---------------------------------------------------------------------------------------------- unsaturated 29-bit form
def test_fqu_field_ops_vs_python(shim):
    """csrc/ffu.cuh: saturated -> U-form -> op -> saturated equals the plain modular result."""
    rng = random.Random(17)
    q = P.Q_MOD
    vals = [0, 1, 2, q - 1, q - 2, (1 << 380) % q] + [rng.randrange(q) for _ in range(120)]
    for a in vals[:12]:
        for b in vals[:12]:
            for op, f in ((0, lambda x, y: x + y), (1, lambda x, y: x - y), (6, lambda x, y: x - y), (2, lambda x, y: x * y)):
                got = P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", op, fq_mont(a), fq_mont(b))))
                assert got == f(a, b) % q, (op, a, b)
    for i in range(0, len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 2, fq_mont(a), fq_mont(b)))) == a * b % q
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 3, fq_mont(a), fq_mont(b)))) == a * a % q
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 5, fq_mont(a), fq_mont(b)))) == (-a) % q
        t = (a * b - ((a + b) - 2 * b)) % q
        u = (t * t - (t * a + 2 * t * b)) % q
        exp = (t * (a - u) - u * b) % q
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 7, fq_mont(a), fq_mont(b)))) == exp
        # fused product-difference with one reduction (fqu_mul2 / f_mul_sub: Y3 of the mixed addition)
        exp9 = ((a - b) * (a + b) - (a * a - a * b - 2 * b * b) * (b - a)) % q
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 9, fq_mont(a), fq_mont(b)))) == exp9


def test_fqu_inverse_vs_python(shim):
    """fqu_inv / Fq2U f_inv / f_tidy (the batched to-affine of the setup's fixed-base multiplication)."""
    rng = random.Random(23)
    q = P.Q_MOD
    for a in [1, 2, q - 1] + [rng.randrange(1, q) for _ in range(6)]:
        b = rng.randrange(1, q)
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 4, fq_mont(a), fq_mont(b)))) == pow(a, q - 2, q)
        assert P.fq_from_mont(unlimbs(call_field(shim, "ht_fqu_op", 8, fq_mont(a), fq_mont(b)))) == pow(a, q - 2, q)
    enc = lambda p: np.concatenate([fq_mont(p.c0), fq_mont(p.c1)])
    dec = lambda a: P.Fq2(P.fq_from_mont(unlimbs(a[:6])), P.fq_from_mont(unlimbs(a[6:])))
    for _ in range(5):
        a = P.Fq2(rng.randrange(q), rng.randrange(1, q))
        b = P.Fq2(rng.randrange(1, q), rng.randrange(q))
        assert dec(call_field(shim, "ht_fq2u_op", 4, enc(a), enc(b))) == a.inv()
        assert dec(call_field(shim, "ht_fq2u_op", 8, enc(a), enc(b))) == a.inv()


def test_fqu_is_zero_mod(shim):
    for k in (0, 1, 2, 3, 7, 33, 63, 64, 74, 100, 127, 148, 1000, 4000):
        assert shim.ht_fqu_is_zero_mod_k(k, 0) == 1, k
        assert shim.ht_fqu_is_zero_mod_k(k, 1) == 0, k
        assert shim.ht_fqu_is_zero_mod_k(k, 12345) == 0, k


def test_fq2u_field_ops_vs_python(shim):
    rng = random.Random(19)
    enc = lambda p: np.concatenate([fq_mont(p.c0), fq_mont(p.c1)])
    dec = lambda a: P.Fq2(P.fq_from_mont(unlimbs(a[:6])), P.fq_from_mont(unlimbs(a[6:])))
    for _ in range(60):
        a = P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD))
        b = P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD))
        assert dec(call_field(shim, "ht_fq2u_op", 2, enc(a), enc(b))) == a * b
        assert dec(call_field(shim, "ht_fq2u_op", 3, enc(a), enc(b))) == a * a
        assert dec(call_field(shim, "ht_fq2u_op", 1, enc(a), enc(b))) == a - b
        t = a * b - ((a + b) - (b + b))
        u = t * t - (t * a + (t * b + t * b))
        assert dec(call_field(shim, "ht_fq2u_op", 7, enc(a), enc(b))) == t * (a - u) - u * b


def pointu_op(shim, group, op, acc, q, neg=0, reps=1):
    w = 12 if group == "g1" else 24
    acc32, q32 = u32(acc), u32(q)
    oa = np.zeros(2 * w, dtype=np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    getattr(shim, "ht_%su_op" % group)(op, p(acc32), p(q32), neg, reps, p(oa))
    return oa.view(np.uint64)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_curve_kat_uform(shim, group):
    """The XYZZ formulas instantiated over the unsaturated types (what the MSM kernels run) vs the golden vectors,
    including the exceptional cases (P + P, P - P, infinity) and chains of un-reduced coordinates."""
    kat = load("curve_kat.json")
    w = 12 if group == "g1" else 24
    gen = G1_GEN_LIMBS if group == "g1" else G2_GEN_LIMBS
    enc = g1_limbs if group == "g1" else g2_limbs
    inf_x = np.zeros(2 * w, dtype=np.uint64)
    gen_x, _ = point_op(shim, group, 4, inf_x, q=gen)
    for v in kat[group + "_add"]:
        ax, aa = point_op(shim, group, 3, gen_x, k=fr_canon(H(v["a"])))      # saturated reference path
        bx, ba = point_op(shim, group, 3, gen_x, k=fr_canon(H(v["b"])))
        exp, _ = enc(v["p"])
        assert np.array_equal(pointu_op(shim, group, 0, ax, ba), exp), v       # U-form mixed add
        assert np.array_equal(pointu_op(shim, group, 1, ax, bx), exp), v       # U-form full add
    # chains: acc = 5G; add G 40 times -> 45G; double 6 times -> 64*G*... vs python
    ax, _ = point_op(shim, group, 3, gen_x, k=fr_canon(5))
    mul = P.g1_mul if group == "g1" else P.g2_mul
    py = py_g1 if group == "g1" else py_g2
    assert np.array_equal(pointu_op(shim, group, 0, ax, gen, reps=40), py(mul(45))[0])
    assert np.array_equal(pointu_op(shim, group, 0, ax, gen, neg=1, reps=3), py(mul(2))[0])
    assert np.array_equal(pointu_op(shim, group, 0, ax, gen, neg=1, reps=5), np.zeros(w, dtype=np.uint64))     # 5G - 5G = inf
    assert np.array_equal(pointu_op(shim, group, 0, ax, gen, neg=1, reps=7), py(mul(P.R_MOD - 2))[0])
    assert np.array_equal(pointu_op(shim, group, 2, ax, gen, reps=9), py(mul(5 * 512))[0])
    assert np.array_equal(pointu_op(shim, group, 1, ax, ax, reps=1), py(mul(10))[0])                          # add of equal points -> doubling branch


def test_fru_ops_vs_python(shim):
    """csrc/fru.cuh (the NTT's unsaturated Fr): saturated -> U-form -> op -> saturated equals the plain modular result,
    including the two conversion-by-multiplication paths the transform uses at load and store."""
    rng = random.Random(29)
    r = P.R_MOD
    vals = [0, 1, 2, r - 1, r - 2, (1 << 254) % r, (1 << 255) % r] + [rng.randrange(r) for _ in range(60)]
    ops = ((0, lambda x, y: x + y), (1, lambda x, y: x - y), (2, lambda x, y: x * y), (3, lambda x, y: (x - y) * y),
           (4, lambda x, y: x * y), (5, lambda x, y: x * y), (6, lambda x, y: (x * y - y) * x * pow(32, -1, P.R_MOD)),
           (7, lambda x, y: x * pow(32, -1, P.R_MOD)),
           (8, lambda x, y: x + 12 * y), (9, lambda x, y: -2 * y * y), (10, lambda x, y: 12 * x - y))       # lazy butterfly sums / raw differences
    for a in vals[:9]:
        for b in vals[:9]:
            for op, f in ops:
                got = P.fr_from_mont(unlimbs(call_field(shim, "ht_fru_op", op, fr_mont(a), fr_mont(b))))
                assert got == f(a, b) % r, (op, a, b)
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        for op, f in ops:
            assert P.fr_from_mont(unlimbs(call_field(shim, "ht_fru_op", op, fr_mont(a), fr_mont(b)))) == f(a, b) % r, (op, a, b)


# ---------------------------------------------------------------------------------------------- host-only 64-bit-limb arithmetic
def _call64(shim, name, op, a, b):
    a64, b64 = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    o = np.zeros_like(a64)
    getattr(shim, name)(op, a64.ctypes.data_as(C.c_void_p), b64.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
    return o


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_h64_fields_vs_python(shim, field):
    """csrc/hostff.hpp (the host chains of the device witness generator and the verifier run on it): add / sub / mul / sqr / inv /
    neg / to_mont / from_mont against Python big ints, on the golden vectors, random values and the edges 0, 1, p - 1."""
    nl, mod, to_m, from_m = (4, P.R_MOD, P.fr_to_mont, P.fr_from_mont) if field == "fr" else (6, P.Q_MOD, P.fq_to_mont, P.fq_from_mont)
    name = "ht_h64_%s_op" % field
    rng = random.Random(64)
    vals = [(H(v["a"]), H(v["b"])) for v in load("field_kat.json")[field]]
    vals += [(rng.randrange(mod), rng.randrange(mod)) for _ in range(200)]
    vals += [(0, 0), (0, 1), (1, mod - 1), (mod - 1, mod - 1), (mod - 1, 1), (2, (mod + 1) // 2)]
    for x, y in vals:
        a, b = limbs(to_m(x), nl), limbs(to_m(y), nl)
        got = lambda op: from_m(unlimbs(_call64(shim, name, op, a, b)))
        assert got(0) == (x + y) % mod and got(1) == (x - y) % mod and got(2) == x * y % mod
        assert got(3) == x * x % mod and got(5) == (-x) % mod
        if x:
            assert got(4) == pow(x, -1, mod)
        assert unlimbs(_call64(shim, name, 6, limbs(x, nl), b)) == to_m(x)
        assert unlimbs(_call64(shim, name, 7, a, b)) == x


def test_verifier_tower_arithmetic_vs_python(shim):
    """csrc/pairing_fast.inc: Fq12 as Fq6[w]/(w^2 - v) over Fq2[v]/(v^3 - xi) on 64-bit limbs — product, square, inverse, the sparse
    line product mul_by_014, Frobenius and conjugation against the flat pure-Python Fq12 (tests/golden/pyref_pairing.py); the
    cyclotomic-subgroup shortcuts (Granger-Scott squaring, conjugate = inverse, the |z| ladder) against the generic arithmetic;
    and the start-up calibration of the endomorphism constants must have succeeded (fast subgroup tests in use)."""
    import pyref_pairing as PP
    rng = random.Random(12)
    tower = [0, 2, 4, 1, 3, 5]                 # ABI slot k holds the coefficient of w^tower[k]

    def rand12():
        return PP.Fq12([P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD)) for _ in range(6)])

    def to_abi(f):
        out = []
        for k in range(6):
            c = f.c[tower[k]]
            out += list(limbs(P.fq_to_mont(c.c0), 6)) + list(limbs(P.fq_to_mont(c.c1), 6))
        return np.array(out, dtype=np.uint64)

    def from_abi(a):
        c = [None] * 6
        for k in range(6):
            c[tower[k]] = P.Fq2(P.fq_from_mont(unlimbs(a[12 * k:12 * k + 6])), P.fq_from_mont(unlimbs(a[12 * k + 6:12 * k + 12])))
        return PP.Fq12(c)

    def call(op, a, b):
        o = np.zeros(72, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        shim.ht_f12_op(op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        return from_abi(o)
    for _ in range(4):
        x, y = rand12(), rand12()
        ax, ay = to_abi(x), to_abi(y)
        assert call(0, ax, ay) == x * y
        assert call(1, ax, ay) == x * x
        assert call(2, ax, ay) * x == PP.Fq12.one()
        # sparse line l0 + l1 w^2 + l4 w^3
        l = [P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD)) for _ in range(3)]
        sparse = PP.Fq12([l[0], P.Fq2(0, 0), l[1], l[2], P.Fq2(0, 0), P.Fq2(0, 0)])
        lb = np.array(sum([list(limbs(P.fq_to_mont(c.c0), 6)) + list(limbs(P.fq_to_mont(c.c1), 6)) for c in l], []) + [0] * 36, dtype=np.uint64)
        assert call(3, ax, lb) == x * sparse
        for times in (1, 2):
            assert call(4, ax, np.array([times] + [0] * 71, dtype=np.uint64)) == x.pow(P.Q_MOD ** times)
        assert call(5, ax, ay) == x.pow(P.Q_MOD ** 6)
        shim.ht_f12_cyclotomic_checks.restype = C.c_int
        assert shim.ht_f12_cyclotomic_checks(ax.ctypes.data_as(C.c_void_p)) == 31
