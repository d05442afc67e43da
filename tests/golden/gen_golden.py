#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the pure-Python big-integer
implementation in pyref.py (NOT from the reference: ArielElb/zkSnark-FinalProject's prover is
Rust + un-vendored arkworks crates and cannot run in this image; it also ships no golden
vectors — SURVEY.md F3/F4).  Run:  python3 tests/golden/gen_golden.py

All integers are written as hex strings of the *canonical* residue; the tests convert to the
Montgomery/limb layout of the C ABI themselves.
"""
import json
import os
import random

import pyref as P

HERE = os.path.dirname(os.path.abspath(__file__))
hx = lambda v: "%x" % v


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def g1j(p):
    return None if p is None else [hx(p[0].v), hx(p[1].v)]


def g2j(p):
    return None if p is None else [hx(p[0].c0), hx(p[0].c1), hx(p[1].c0), hx(p[1].c1)]


def gen_constants():
    dump("constants.json", {
        "fr_modulus": hx(P.R_MOD), "fq_modulus": hx(P.Q_MOD),
        "fr_mont_r": hx(P.FR_MONT_R), "fq_mont_r": hx(P.FQ_MONT_R),
        "fr_mont_r2": hx(pow(2, 512, P.R_MOD)), "fq_mont_r2": hx(pow(2, 768, P.Q_MOD)),
        "fr_inv64": hx((-pow(P.R_MOD, -1, 1 << 64)) % (1 << 64)),
        "fq_inv64": hx((-pow(P.Q_MOD, -1, 1 << 64)) % (1 << 64)),
        "fr_root_2_32": hx(P.FR_ROOT_2_32), "fr_generator": hx(P.FR_GEN),
        "roots": {str(k): hx(P.root_of_unity(k)) for k in range(0, 33)},
        "g1_gen": g1j(P.G1_GEN), "g2_gen": g2j(P.G2_GEN),
    })


def gen_field(rng):
    out = {"fr": [], "fq": [], "fq2": []}
    for mod, key in ((P.R_MOD, "fr"), (P.Q_MOD, "fq")):
        edge = [0, 1, 2, mod - 1, mod - 2, (1 << 64) - 1, 1 << 64, (1 << 255) % mod]
        pairs = [(a, b) for a in edge[:5] for b in edge[:5]] + \
                [(rng.randrange(mod), rng.randrange(mod)) for _ in range(40)] + \
                [(e, rng.randrange(mod)) for e in edge]
        for a, b in pairs:
            out[key].append({"a": hx(a), "b": hx(b), "add": hx((a + b) % mod), "sub": hx((a - b) % mod),
                             "mul": hx(a * b % mod), "inv_a": hx(pow(a, -1, mod) if a else 0)})
    for _ in range(24):
        a = P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD))
        b = P.Fq2(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD))
        m, s, i = a * b, a * a, a.inv()
        out["fq2"].append({"a": [hx(a.c0), hx(a.c1)], "b": [hx(b.c0), hx(b.c1)], "mul": [hx(m.c0), hx(m.c1)],
                           "sqr": [hx(s.c0), hx(s.c1)], "inv_a": [hx(i.c0), hx(i.c1)]})
    dump("field_kat.json", out)


def gen_curve(rng):
    ks = [1, 2, 3, 5, 0xffff, P.R_MOD - 1, P.R_MOD - 2, (1 << 128) + 12345] + [P.rand_fr(rng) for _ in range(8)]
    out = {"g1_mul": [], "g2_mul": [], "g1_add": [], "g2_add": []}
    for k in ks:
        out["g1_mul"].append({"k": hx(k), "p": g1j(P.g1_mul(k))})
        out["g2_mul"].append({"k": hx(k), "p": g2j(P.g2_mul(k))})
    for _ in range(6):
        a, b = P.rand_fr(rng), P.rand_fr(rng)
        out["g1_add"].append({"a": hx(a), "b": hx(b), "p": g1j(P.g1_mul(a + b))})
        out["g2_add"].append({"a": hx(a), "b": hx(b), "p": g2j(P.g2_mul(a + b))})
    # exceptional cases: P+P, P+(-P), P+inf
    for a, b in ((7, 7), (7, P.R_MOD - 7), (9, 0), (0, 9), (0, 0)):
        out["g1_add"].append({"a": hx(a), "b": hx(b), "p": g1j(P.g1_mul(a + b))})
        out["g2_add"].append({"a": hx(a), "b": hx(b), "p": g2j(P.g2_mul(a + b))})
    dump("curve_kat.json", out)


def gen_ntt(rng):
    out = []
    for log_n in (0, 1, 2, 3, 5, 7):
        n = 1 << log_n
        a = [P.rand_fr(rng) for _ in range(n)]
        if log_n == 3:
            a[0] = 0
            a[1] = P.R_MOD - 1
        case = {"log_n": log_n, "in": [hx(x) for x in a]}
        for inv in (0, 1):
            for coset in (0, 1):
                case["out_inv%d_coset%d" % (inv, coset)] = [hx(x) for x in P.dft_naive(a, bool(inv), bool(coset))]
        out.append(case)
    dump("ntt_kat.json", out)


def gen_msm(rng):
    """bases are [k_i]G with known k_i so the expected result is [sum s_i k_i]G."""
    out = []
    specs = [("empty", 0), ("one", 1), ("two", 2), ("seven", 7), ("n31", 31), ("n32", 32), ("n33", 33), ("n100", 100)]
    for name, n in specs:
        ks = [P.rand_fr(rng) or 1 for _ in range(n)]
        ss = [P.rand_fr(rng) for _ in range(n)]
        inf = [0] * n
        if n >= 7:
            ss[0] = 0
            ss[1] = 1
            ss[2] = P.R_MOD - 1
            ss[3] = (1 << 255) % P.R_MOD
            ks[5] = ks[4]                       # repeated base
            ks[6] = (P.R_MOD - ks[4]) % P.R_MOD  # and its negative
        if n >= 31:
            for c in (3, 7, 8, 9, 15, 16):       # window-boundary digits for several c
                ss[7 + c % 20] = (1 << (c - 1))
                ss[8 + c % 19] = (1 << c) - 1
            inf[10] = 1                          # infinity base (affine flag) with a non-zero scalar
            inf[11] = 1
            ss[12] = ss[13] = 5                  # equal scalars: same bucket in every window
            ks[13] = ks[12]                      # ... on the same point (forces a doubling in the bucket)
        total = sum(s * k for s, k, f in zip(ss, ks, inf) if not f) % P.R_MOD
        out.append({"name": name, "n": n, "base_logs": [hx(k) for k in ks], "inf": inf,
                    "scalars": [hx(s) for s in ss], "expected_log": hx(total),
                    "expected_g1": g1j(P.g1_mul(total)), "expected_g2": g2j(P.g2_mul(total))})
    # all-ones / all-zero / skewed (the matmul witness shape: SURVEY.md 8d)
    n = 64
    ks = [P.rand_fr(rng) for _ in range(n)]
    for name, ss in (("all_ones", [1] * n), ("all_zero", [0] * n), ("all_same", [0x1234567] * n),
                     ("bits", [rng.randrange(2) for _ in range(n)])):
        total = sum(s * k for s, k in zip(ss, ks)) % P.R_MOD
        out.append({"name": name, "n": n, "base_logs": [hx(k) for k in ks], "inf": [0] * n,
                    "scalars": [hx(s) for s in ss], "expected_log": hx(total),
                    "expected_g1": g1j(P.g1_mul(total)), "expected_g2": g2j(P.g2_mul(total))})
    dump("msm_kat.json", out)


def random_r1cs(rng, nc, num_inputs, num_vars, max_nnz=4):
    """A satisfiable random R1CS: choose A, B rows and z, then solve each C row with one fresh-ish term."""
    z = [1] + [P.rand_fr(rng) for _ in range(num_vars - 1)]
    z[2 % num_vars] = 0 if num_vars > 2 else z[2 % num_vars]
    A, B, Cm = [], [], []
    for i in range(nc):
        ra = [(P.rand_fr(rng) if rng.random() < .5 else rng.randrange(1, 4), rng.randrange(num_vars))
              for _ in range(rng.randrange(1, max_nnz + 1))]
        rb = [(P.rand_fr(rng) if rng.random() < .5 else 1, rng.randrange(num_vars))
              for _ in range(rng.randrange(1, max_nnz + 1))]
        if i % 7 == 3:
            rb = []                                   # empty row: <B_i, z> = 0
        av = sum(c * z[j] for c, j in ra) % P.R_MOD
        bv = sum(c * z[j] for c, j in rb) % P.R_MOD
        target = av * bv % P.R_MOD
        # C row: a random term plus a correcting multiple of the constant 1 (column 0) -> satisfied
        j = rng.randrange(1, num_vars)
        c1 = P.rand_fr(rng)
        rc = [(c1, j), ((target - c1 * z[j]) % P.R_MOD, 0)]
        if target == 0 and i % 7 == 3:
            rc = []
        A.append(ra)
        B.append(rb)
        Cm.append(rc)
    return A, B, Cm, z


def gen_groth16(rng):
    out = []
    cubic = ([[(1, 2)], [(1, 3)], [(1, 4), (1, 2), (5, 0)]],
             [[(1, 2)], [(1, 2)], [(1, 0)]],
             [[(1, 3)], [(1, 4)], [(1, 1)]], [1, 35, 3, 9, 27], 2)
    A, B, Cm, z = random_r1cs(rng, 27, 3, 14)
    for name, (A, B, Cm, z, ni) in (("cubic", cubic), ("random27", (A, B, Cm, z, 3))):
        nv = len(z)
        trap = {k: P.rand_fr(rng) for k in ("tau", "alpha", "beta", "gamma", "delta")}
        g1s, g2s = P.rand_fr(rng), P.rand_fr(rng)       # arkworks' setup uses random generators
        r, s = P.rand_fr(rng), P.rand_fr(rng)
        N, h = P.witness_map_h(A, B, Cm, ni, z)
        logs = P.groth16_setup_logs(A, B, Cm, ni, nv, trap)
        a, b, c, ok = P.groth16_prove_logs(logs, h, z, ni, r, s)
        assert ok, "in-the-exponent Groth16 check failed"
        g1 = lambda k: g1j(P.g1_mul(k * g1s))
        g2 = lambda k: g2j(P.g2_mul(k * g2s))
        out.append({
            "name": name, "num_inputs": ni, "num_vars": nv, "num_constraints": len(A), "N": N,
            "A": [[[hx(cf), j] for cf, j in row] for row in A],
            "B": [[[hx(cf), j] for cf, j in row] for row in B],
            "C": [[[hx(cf), j] for cf, j in row] for row in Cm],
            "z": [hx(v) for v in z], "trapdoor": {k: hx(v) for k, v in trap.items()},
            "g1_scalar": hx(g1s), "g2_scalar": hx(g2s), "r": hx(r), "s": hx(s),
            "h": [hx(v) for v in h],
            "logs": {k: ([hx(v) for v in logs[k]]) for k in ("a_query", "b_query", "h_query", "l_query", "gamma_abc")},
            "pk": {
                "a_query": [g1(k) for k in logs["a_query"]], "b_g1_query": [g1(k) for k in logs["b_query"]],
                "b_g2_query": [g2(k) for k in logs["b_query"]], "h_query": [g1(k) for k in logs["h_query"]],
                "l_query": [g1(k) for k in logs["l_query"]],
                "alpha_g1": g1(trap["alpha"]), "beta_g1": g1(trap["beta"]), "beta_g2": g2(trap["beta"]),
                "delta_g1": g1(trap["delta"]), "delta_g2": g2(trap["delta"]),
            },
            "proof_logs": {"a": hx(a), "b": hx(b), "c": hx(c)},
            "proof": {"a": g1(a), "b": g2(b), "c": g1(c)},
        })
    dump("groth16_kat.json", out)


if __name__ == "__main__":
    rng = random.Random(0x5EED2026)
    gen_constants()
    gen_field(rng)
    gen_curve(rng)
    gen_ntt(rng)
    gen_msm(rng)
    gen_groth16(rng)
