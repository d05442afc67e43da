"""Pins the CPU oracle (oracle/g16_oracle.c) against the golden fixtures generated from the
independent pure-Python big-integer implementation (tests/golden/gen_golden.py).
The reference holds no vectors for this path — parity with arkworks bytes is
unpinned; these fixtures are what stands in."""
import numpy as np
import pytest

import pyref as P
from helpers import *


def test_constants(oracle):
    c = load("constants.json")
    assert H(c["fr_modulus"]) == P.R_MOD and H(c["fq_modulus"]) == P.Q_MOD
    one = oracle.fr_from_canonical(limbs(1, 4))
    assert unlimbs(one) == H(c["fr_mont_r"])
    one = oracle.fq_from_canonical(limbs(1, 6))
    assert unlimbs(one) == H(c["fq_mont_r"])
    # roots of unity: w_k has exact order 2^k
    for k, v in c["roots"].items():
        w = H(v)
        assert pow(w, 1 << int(k), P.R_MOD) == 1
        if int(k) > 0:
            assert pow(w, 1 << (int(k) - 1), P.R_MOD) == P.R_MOD - 1
    g1, _ = g1_limbs(c["g1_gen"])
    g2, _ = g2_limbs(c["g2_gen"])
    assert oracle.lib().orc_g1_on_curve(g1) == 1
    assert oracle.lib().orc_g2_on_curve(g2) == 1


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_field_kat(oracle, field):
    nl, to_m, from_m = (4, P.fr_to_mont, P.fr_from_mont) if field == "fr" else (6, P.fq_to_mont, P.fq_from_mont)
    for v in load("field_kat.json")[field]:
        a, b = limbs(to_m(H(v["a"])), nl), limbs(to_m(H(v["b"])), nl)
        for op in ("add", "sub", "mul"):
            got = from_m(unlimbs(oracle.binop("%s_%s" % (field, op), a, b)))
            assert got == H(v[op]), (field, op, v)
        assert from_m(unlimbs(oracle.unop(field + "_inv", a))) == H(v["inv_a"])
        # canonical <-> Montgomery converters
        conv = getattr(oracle, field + "_from_canonical")(limbs(H(v["a"]), nl))
        assert unlimbs(conv) == to_m(H(v["a"]))
        back = getattr(oracle, field + "_to_canonical")(conv)
        assert unlimbs(back) == H(v["a"])


def test_fq2_kat(oracle):
    enc = lambda p: np.concatenate([fq_mont(H(p[0])), fq_mont(H(p[1]))])
    dec = lambda a: [P.fq_from_mont(unlimbs(a[:6])), P.fq_from_mont(unlimbs(a[6:]))]
    for v in load("field_kat.json")["fq2"]:
        a, b = enc(v["a"]), enc(v["b"])
        assert dec(oracle.binop("fq2_mul", a, b)) == [H(x) for x in v["mul"]]
        assert dec(oracle.unop("fq2_sqr", a)) == [H(x) for x in v["sqr"]]
        assert dec(oracle.unop("fq2_inv", a)) == [H(x) for x in v["inv_a"]]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_curve_kat(oracle, group):
    kat = load("curve_kat.json")
    gen = G1_GEN_LIMBS if group == "g1" else G2_GEN_LIMBS
    enc = g1_limbs if group == "g1" else g2_limbs
    for v in kat[group + "_mul"]:
        got, inf = oracle.point_mul(group, gen, fr_canon(H(v["k"])))
        exp, einf = enc(v["p"])
        assert inf == einf and (einf or np.array_equal(got, exp)), v["k"]
    for v in kat[group + "_add"]:
        pa, ia = oracle.point_mul(group, gen, fr_canon(H(v["a"])))
        pb, ib = oracle.point_mul(group, gen, fr_canon(H(v["b"])))
        got, inf = oracle.point_add(group, pa, pb, ia, ib)
        exp, einf = enc(v["p"])
        assert inf == einf and (einf or np.array_equal(got, exp)), v


def test_ntt_kat(oracle):
    for case in load("ntt_kat.json"):
        a = fr_mont_vec([H(x) for x in case["in"]])
        for inv in (0, 1):
            for coset in (0, 1):
                got = fr_from_mont_vec(oracle.ntt(a, bool(inv), bool(coset)))
                assert got == [H(x) for x in case["out_inv%d_coset%d" % (inv, coset)]], (case["log_n"], inv, coset)


def test_ntt_roundtrip_and_horner(oracle):
    """N = 2^12: ifft(fft(x)) == x, and fft output equals Horner evaluation at random domain points."""
    import random
    rng = random.Random(7)
    n = 1 << 12
    xs = [P.rand_fr(rng) for _ in range(n)]
    a = fr_mont_vec(xs)
    f = oracle.ntt(a)
    assert fr_from_mont_vec(oracle.ntt(f, inverse=True)) == xs
    fc = oracle.ntt(a, coset=True)
    assert fr_from_mont_vec(oracle.ntt(fc, inverse=True, coset=True)) == xs
    w = P.root_of_unity(12)
    fv, fcv = fr_from_mont_vec(f), fr_from_mont_vec(fc)
    for k in (0, 1, 2, 1234, n - 1):
        assert fv[k] == P.poly_eval(xs, pow(w, k, P.R_MOD))
        assert fcv[k] == P.poly_eval(xs, P.FR_GEN * pow(w, k, P.R_MOD) % P.R_MOD)


def _msm_case_arrays(case, group, oracle):
    gen = G1_GEN_LIMBS if group == "g1" else G2_GEN_LIMBS
    logs = fr_canon_vec([H(k) for k in case["base_logs"]])
    if case["n"]:
        bases, binf = oracle.fixed_base(group, gen, logs)
    else:
        bases, binf = np.zeros((0, 12 if group == "g1" else 24), np.uint64), np.zeros(0, np.uint8)
    inf = np.array(case["inf"], dtype=np.uint8) | binf
    return bases, inf, fr_canon_vec([H(s) for s in case["scalars"]])


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_kat(oracle, group):
    enc = g1_limbs if group == "g1" else g2_limbs
    for case in load("msm_kat.json"):
        bases, inf, sc = _msm_case_arrays(case, group, oracle)
        got, ginf = oracle.msm(group, bases, sc, inf)
        exp, einf = enc(case["expected_" + group])
        assert ginf == einf and (einf or np.array_equal(got, exp)), case["name"]


def test_fixed_base_matches_double_and_add(oracle):
    import random
    rng = random.Random(3)
    ks = [0, 1, 255, 256, P.R_MOD - 1] + [P.rand_fr(rng) for _ in range(5)]
    for group, gen, py in (("g1", G1_GEN_LIMBS, lambda k: py_g1(P.g1_mul(k))), ("g2", G2_GEN_LIMBS, lambda k: py_g2(P.g2_mul(k)))):
        pts, inf = oracle.fixed_base(group, gen, fr_canon_vec(ks))
        for i, k in enumerate(ks):
            exp, einf = py(k)
            assert inf[i] == einf and (einf or np.array_equal(pts[i], exp))


def test_witness_map_and_setup_logs(oracle):
    for case in load("groth16_kat.json"):
        r1cs, _ = r1cs_from_case(case)
        z = fr_mont_vec([H(v) for v in case["z"]])
        h = oracle.witness_map(r1cs, z)
        assert fr_from_mont_vec(h) == [H(v) for v in case["h"]], case["name"]
        t = case["trapdoor"]
        trap = fr_mont_vec([H(t[k]) for k in ("tau", "alpha", "beta", "gamma", "delta")])
        logs = oracle.setup_logs(r1cs, case["num_vars"], trap)
        assert logs["N"] == case["N"]
        for ours, theirs in (("a", "a_query"), ("b", "b_query"), ("l", "l_query"), ("h", "h_query"), ("gabc", "gamma_abc")):
            assert fr_from_mont_vec(logs[ours]) == [H(v) for v in case["logs"][theirs]], (case["name"], ours)


def test_prove_kat(oracle):
    """Whole proofs: oracle (real MSMs + NTTs) == Python proof computed in the exponent."""
    for case in load("groth16_kat.json"):
        r1cs, _ = r1cs_from_case(case)
        pk = pk_from_case(case)
        z = fr_mont_vec([H(v) for v in case["z"]])
        proof, inf = oracle.prove(pk, fr_mont(H(case["r"])), fr_mont(H(case["s"])), r1cs, z)
        assert list(inf) == [0, 0, 0]
        assert np.array_equal(proof[:12], g1_limbs(case["proof"]["a"])[0]), case["name"]
        assert np.array_equal(proof[12:36], g2_limbs(case["proof"]["b"])[0]), case["name"]
        assert np.array_equal(proof[36:], g1_limbs(case["proof"]["c"])[0]), case["name"]
