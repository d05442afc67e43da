"""Shared test helpers: golden-fixture loading and conversions between Python ints and the
C-ABI limb layout (Montgomery, little-endian u64 limbs — include/zkg16.h)."""
import json
import os

import numpy as np

import pyref as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK64 = (1 << 64) - 1


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def H(s):
    return int(s, 16)


def limbs(x, n):
    return np.array([(x >> (64 * i)) & MASK64 for i in range(n)], dtype=np.uint64)


def unlimbs(a):
    return sum(int(v) << (64 * i) for i, v in enumerate(np.asarray(a).ravel()))


def fr_mont(x):
    return limbs(P.fr_to_mont(x % P.R_MOD), 4)


def fr_canon(x):
    return limbs(x % P.R_MOD, 4)


def fq_mont(x):
    return limbs(P.fq_to_mont(x % P.Q_MOD), 6)


def fr_mont_vec(xs):
    return np.array([fr_mont(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def fr_canon_vec(xs):
    return np.array([fr_canon(x) for x in xs], dtype=np.uint64).reshape(-1, 4)


def fr_from_mont_vec(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [P.fr_from_mont(unlimbs(r)) for r in a]


def g1_limbs(j):
    """golden json point ([x,y] hex or None) -> (12 limbs, inf)"""
    if j is None:
        return np.zeros(12, dtype=np.uint64), 1
    return np.concatenate([fq_mont(H(j[0])), fq_mont(H(j[1]))]), 0


def g2_limbs(j):
    if j is None:
        return np.zeros(24, dtype=np.uint64), 1
    return np.concatenate([fq_mont(H(v)) for v in j]), 0


def g1_vec(js):
    pts = [g1_limbs(j) for j in js]
    return (np.array([p for p, _ in pts], dtype=np.uint64).reshape(-1, 12),
            np.array([f for _, f in pts], dtype=np.uint8))


def g2_vec(js):
    pts = [g2_limbs(j) for j in js]
    return (np.array([p for p, _ in pts], dtype=np.uint64).reshape(-1, 24),
            np.array([f for _, f in pts], dtype=np.uint8))


def py_g1(p):
    return g1_limbs(None if p is None else ["%x" % p[0].v, "%x" % p[1].v])


def py_g2(p):
    return g2_limbs(None if p is None else ["%x" % p[0].c0, "%x" % p[0].c1, "%x" % p[1].c0, "%x" % p[1].c1])


G1_GEN_LIMBS = py_g1(P.G1_GEN)[0]
G2_GEN_LIMBS = py_g2(P.G2_GEN)[0]


def csr_from_rows(rows, ncols=None):
    """rows: list of [(coeff_int, col)] -> (row_ptr u64, col u32, coeff (nnz,4) Montgomery)"""
    rp = [0]
    col = []
    cf = []
    for row in rows:
        for c, j in row:
            col.append(j)
            cf.append(fr_mont(c))
        rp.append(len(col))
    cfa = np.array(cf, dtype=np.uint64).reshape(-1, 4) if cf else np.zeros((0, 4), dtype=np.uint64)
    return np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32), cfa


def r1cs_from_case(case):
    A = [[(H(c), j) for c, j in row] for row in case["A"]]
    B = [[(H(c), j) for c, j in row] for row in case["B"]]
    C = [[(H(c), j) for c, j in row] for row in case["C"]]
    return dict(a=csr_from_rows(A), b=csr_from_rows(B), c=csr_from_rows(C),
                num_inputs=case["num_inputs"], num_constraints=case["num_constraints"]), (A, B, C)


def pk_from_case(case):
    pk = {}
    for k in ("a_query", "b_g1_query", "h_query", "l_query"):
        pk[k], pk[k.replace("_query", "") + "_inf" if k != "b_g1_query" else "b_g1_inf"] = g1_vec(case["pk"][k])
    pk["b_g2_query"], pk["b_g2_inf"] = g2_vec(case["pk"]["b_g2_query"])
    for k in ("alpha_g1", "beta_g1", "delta_g1"):
        pk[k] = g1_limbs(case["pk"][k])[0]
    for k in ("beta_g2", "delta_g2"):
        pk[k] = g2_limbs(case["pk"][k])[0]
    return pk
