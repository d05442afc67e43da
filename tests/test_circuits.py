"""Host-side circuit synthesis (csrc/circuits.hip) — the C++ mirrors of the reference's MatrixCircuit (+ Poseidon sponge)
and FibonacciCircuit.  Checked against SURVEY.md Appendix B's counts (derived from the reference source), an independent
pure-Python Poseidon over the same parameter data, R1CS satisfaction, and the oracle's witness map."""
import json
import os
import random

import numpy as np
import pytest

import pyref as P
from helpers import *

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = json.load(open(os.path.join(ROOT, "zksnark-finalproject_amd", "poseidon_bls381_params.json")))
R = P.R_MOD


def py_permute(st):
    mds = [[int(x) for x in row] for row in PARAMS["mds"]]
    ark = [[int(x) for x in row] for row in PARAMS["ark"]]
    half = PARAMS["full_rounds"] // 2
    for r in range(PARAMS["full_rounds"] + PARAMS["partial_rounds"]):
        st = [(s + ark[r][i]) % R for i, s in enumerate(st)]
        if r < half or r >= half + PARAMS["partial_rounds"]:
            st = [pow(s, 17, R) for s in st]
        else:
            st[0] = pow(st[0], 17, R)
        st = [sum(st[j] * mds[i][j] for j in range(3)) % R for i in range(3)]
    return st


def py_poseidon(elems):
    """ark-crypto-primitives PoseidonSponge: absorb all, squeeze 1 (rate 2, capacity 1)."""
    st, pos = [0, 0, 0], 0
    for e in elems:
        if pos == 2:
            st, pos = py_permute(st), 0
        st[1 + pos] = (st[1 + pos] + e) % R
        pos += 1
    return py_permute(st)[1]


def test_poseidon_native_matches_python():
    from zksnark_finalproject_amd.circuits import poseidon_hash
    rng = random.Random(1)
    for n in (1, 2, 3, 4, 9, 16):
        xs = [P.rand_fr(rng) for _ in range(n)]
        got = P.fr_from_mont(unlimbs(poseidon_hash(fr_mont_vec(xs))))
        assert got == py_poseidon(xs), n
    assert P.fr_from_mont(unlimbs(poseidon_hash(fr_mont_vec([1] * 4)))) == py_poseidon([1, 1, 1, 1])


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8])
def test_matrix_circuit_shape_and_values(n, oracle):
    from zksnark_finalproject_amd.circuits import matrix_circuit
    from zksnark_finalproject_amd.workloads import matmul_shape
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1 << 20, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 20, size=(n, n), dtype=np.uint64)
    c = matrix_circuit(a, b)
    shp = matmul_shape(n)
    assert (c.num_instance, c.num_witness, c.num_constraints) == (4, shp["num_witness"], shp["nc"])
    assert c.satisfied
    # public inputs = native Poseidon hashes of A, B and C = A*B (matrix_proof.rs:104-125, :202)
    ai, bi = [[int(x) for x in row] for row in a], [[int(x) for x in row] for row in b]
    ci = [[sum(ai[i][k] * bi[k][j] for k in range(n)) % R for j in range(n)] for i in range(n)]
    flat = lambda m: [x for row in m for x in row]
    pub = fr_from_mont_vec(c.public_inputs)
    assert pub == [py_poseidon(flat(ai)), py_poseidon(flat(bi)), py_poseidon(flat(ci))]
    # allocation order: instance [1, hash_a, hash_b, hash_c]; witnesses start with A then B row-major (constraints.rs:106-109)
    z = fr_from_mont_vec(c.z)
    assert z[0] == 1 and z[4:4 + n * n] == flat(ai) and z[4 + n * n:4 + 2 * n * n] == flat(bi)
    # the QAP quotient exists: deg h <= N - 2
    h = oracle.witness_map(c.r1cs, c.z)
    assert not h[-1].any()
    # every column index is in range; C rows of the product constraints carry one term
    for m in "abc":
        assert c.r1cs[m][1].max() < c.num_vars


def test_matrix_circuit_all_ones_witness_mix():
    """bench/matrix.py:11 inputs: 2n^2 + n^3 ones, 2n^2 zeros."""
    from zksnark_finalproject_amd.circuits import matrix_circuit
    n = 6
    c = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
    z = fr_from_mont_vec(c.z[4:])
    assert sum(1 for v in z if v == 1) >= 2 * n * n + n ** 3
    assert sum(1 for v in z if v == 0) == 2 * n * n


@pytest.mark.parametrize("steps", [0, 1, 10, 186, 1000])
def test_fibonacci_circuit(steps, oracle):
    from zksnark_finalproject_amd.circuits import fibonacci_circuit
    c = fibonacci_circuit(0, 1, steps)
    assert (c.num_instance, c.num_witness, c.num_constraints) == (4, 1, steps + 1)       # fibbonaci.rs:22-48
    assert c.satisfied
    x, y, res = 0, 1, 0
    for _ in range(steps):
        res = (x + y) % R
        x, y = y, res
    assert fr_from_mont_vec(c.public_inputs) == [0, 1, res]
    h = oracle.witness_map(c.r1cs, c.z)
    assert not h.any() or not h[-1].any()


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_poseidon_template_replay_equals_generic_synthesis(n, monkeypatch):
    """permute_gadget replays later Poseidon permutations from a row template (csrc/circuits.hip); the result must be the
    plain gate-by-gate synthesis exactly: same CSR arrays for A, B, C and the same assignment (odd n: ragged last block)."""
    from zksnark_finalproject_amd.circuits import matrix_circuit
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1 << 20, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 20, size=(n, n), dtype=np.uint64)
    fast = matrix_circuit(a, b)
    monkeypatch.setenv("ZKG16_SYNTH_GENERIC", "1")
    plain = matrix_circuit(a, b)
    assert fast.num_constraints == plain.num_constraints and fast.num_vars == plain.num_vars
    for m in ("a", "b", "c"):
        for x, y in zip(fast.r1cs[m], plain.r1cs[m]):
            assert np.array_equal(x, y), m
    assert np.array_equal(fast.z, plain.z)


@pytest.mark.parametrize("n", [1, 2, 7])
def test_segmented_threaded_build_equals_in_order_build(n, monkeypatch):
    """zkg16_circuit_matrix builds the hash_a / hash_b gadgets on their own threads into segments placed by a predicted
    witness offset (csrc/circuits.hip); in-order building (ZKG16_SYNTH_THREADS=0) must give the same CSR arrays and assignment.
    n = 1 exercises the path where the prediction does not apply."""
    from zksnark_finalproject_amd.circuits import matrix_circuit
    rng = np.random.default_rng(100 + n)
    a = rng.integers(0, 1 << 30, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 30, size=(n, n), dtype=np.uint64)
    fast = matrix_circuit(a, b)
    monkeypatch.setenv("ZKG16_SYNTH_THREADS", "0")
    plain = matrix_circuit(a, b)
    assert fast.num_constraints == plain.num_constraints and fast.num_vars == plain.num_vars
    for m in ("a", "b", "c"):
        for x, y in zip(fast.r1cs[m], plain.r1cs[m]):
            assert np.array_equal(x, y), m
    assert np.array_equal(fast.z, plain.z)


@pytest.mark.parametrize("n,chunks", [(16, 2), (17, 4), (23, 3), (24, 4), (24, 16), (34, 4), (35, 1)])
def test_chunked_sponges_equal_in_order_build(n, chunks, monkeypatch):
    """From 128 permutations on, every Poseidon sponge of the MatrixCircuit is built as chunks of permutations on their own
    threads: a chunk gets the native state in front of the permutation before its first one and replays that permutation on a
    scratch circuit to obtain its entering state (csrc/circuits.hip, plan_sponge_chunks).  Same CSR arrays, same assignment and
    the same hashes as the in-order build; odd n: the last permutation absorbs one element; from n = 32 on matrix_mul is built as
    three segments of product rows as well."""
    from zksnark_finalproject_amd.circuits import matrix_circuit
    rng = np.random.default_rng(900 + n)
    a = rng.integers(0, 1 << 40, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 40, size=(n, n), dtype=np.uint64)
    monkeypatch.setenv("ZKG16_SYNTH_CHUNKS", str(chunks))
    monkeypatch.setenv("ZKG16_SYNTH_STRICT", "1")        # no silent fall-back to the in-order build
    fast = matrix_circuit(a, b)
    monkeypatch.setenv("ZKG16_SYNTH_THREADS", "0")
    plain = matrix_circuit(a, b)
    assert fast.num_constraints == plain.num_constraints and fast.num_vars == plain.num_vars
    for m in ("a", "b", "c"):
        for x, y in zip(fast.r1cs[m], plain.r1cs[m]):
            assert np.array_equal(x, y), m
    assert np.array_equal(fast.z, plain.z)
    assert np.array_equal(fast.public_inputs, plain.public_inputs)
    assert fast.satisfied in (True, None) and plain.satisfied in (True, None)      # None: not evaluated above 200,000 constraints


@pytest.mark.parametrize("n", [2, 5, 12, 33])
def test_assignment_only_build_equals_full_synthesis(n):
    """zkg16_circuit_matrix_witness (term bookkeeping off in every builder thread) gives exactly the assignment of the full
    synthesis; a wrong variable count is rejected."""
    from zksnark_finalproject_amd.circuits import matrix_circuit, matrix_witness
    from zksnark_finalproject_amd import Zkg16Error
    rng = np.random.default_rng(500 + n)
    a = rng.integers(0, 1 << 50, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 50, size=(n, n), dtype=np.uint64)
    c = matrix_circuit(a, b)
    assert np.array_equal(matrix_witness(a, b, c.num_vars), c.z)
    with pytest.raises(Zkg16Error):
        matrix_witness(a, b, c.num_vars + 1)


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_matrix_sponge_states_host_half_of_the_device_witness(n):
    """zkg16_matrix_sponge_states (csrc/witness.hip, 64-bit-limb chains with lazily reduced linear layers) against the
    pure-Python sponge: every recorded entering state, the three hashes, and c = a b with entries that overflow 64 bits."""
    from zksnark_finalproject_amd.circuits import matrix_sponge_states, matrix_witness
    from zksnark_finalproject_amd.workloads import matmul_shape
    rng = np.random.default_rng(100 + n)
    a = rng.integers(0, 1 << 63, size=(n, n), dtype=np.uint64) * np.uint64(2) + np.uint64(1)       # full 64-bit entries
    b = rng.integers(0, 1 << 63, size=(n, n), dtype=np.uint64) * np.uint64(2)
    states, hashes = matrix_sponge_states(a, b)
    ai, bi = [[int(x) for x in row] for row in a], [[int(x) for x in row] for row in b]
    ci = [[sum(ai[i][k] * bi[k][j] for k in range(n)) % R for j in range(n)] for i in range(n)]
    flat = lambda m: [x for row in m for x in row]
    for h, elems in enumerate((flat(ai), flat(bi), flat(ci))):
        st, want = [0, 0, 0], []
        for p in range((n * n + 1) // 2):
            for pos, e in enumerate(elems[2 * p:2 * p + 2]):
                st[1 + pos] = (st[1 + pos] + e) % R
            want.append(list(st))
            st = py_permute(st)
        got = [fr_from_mont_vec(states[h, p]) for p in range(len(want))]
        assert got == want, h
        assert P.fr_from_mont(unlimbs(hashes[h])) == st[1]
    # and the same hashes as the host-side assignment builder puts into z[1..3]
    z = matrix_witness(a, b, matmul_shape(n)["num_witness"] + 4)
    assert np.array_equal(z[1:4], hashes)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 9, 12, 17])
def test_matrix_r1cs_from_plan_equals_full_synthesis(n):
    """The MatrixCircuit's R1CS written from its plan (one template per Poseidon-permutation class with symbolic slots + the closed
    form of matrix_mul's rows: csrc/matrix_plan.hpp) == the R1CS of the full gadget-level synthesis, array for array; even and odd
    n (an odd n^2 ends each sponge with a single-element block: its own template class)."""
    from zksnark_finalproject_amd.circuits import matrix_circuit, matrix_r1cs_from_plan
    ones = np.ones((n, n), dtype=np.uint64)
    full = matrix_circuit(ones, ones)
    plan, nw = matrix_r1cs_from_plan(n)
    assert nw == full.num_witness and plan["num_constraints"] == full.num_constraints
    for m in "abc":
        for k, name in enumerate(("row_ptr", "col", "coeff")):
            assert np.array_equal(plan[m][k], full.r1cs[m][k]), (m, name)
