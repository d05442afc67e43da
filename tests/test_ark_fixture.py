"""Off-box parity hook: fixtures dumped by the real ark-groth16 prover with the `dump_fixture` example of
INTEGRATION.md section 6, dropped as tests/golden/ark_fixture_*.bin, are proved again by the oracle (CPU) and by the HIP path
(GPU) and must come out byte-identical.  No such file ships (no cargo in this image: parity vs arkworks bytes stays unpinned);
the format itself is exercised with a fixture written from this repo's own golden case."""
import glob
import os

import numpy as np
import pytest

import ark_fixture
from helpers import *

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ark_fixture_*.bin")))


def _own_fixture(tmp_path):
    case = load("groth16_kat.json")[0]
    r1cs, _ = r1cs_from_case(case)
    proof = np.concatenate([g1_limbs(case["proof"]["a"])[0], g2_limbs(case["proof"]["b"])[0], g1_limbs(case["proof"]["c"])[0]])
    fx = dict(r=fr_mont(H(case["r"])), s=fr_mont(H(case["s"])), r1cs=r1cs, z=fr_mont_vec([H(v) for v in case["z"]]),
              pk=pk_from_case(case), proof=proof, inf=[0, 0, 0])
    path = os.path.join(tmp_path, "ark_fixture_selftest.bin")
    ark_fixture.dump(path, fx)
    return path, fx


def test_fixture_format_roundtrip_and_oracle(tmp_path, oracle):
    path, fx = _own_fixture(str(tmp_path))
    back = ark_fixture.load(path)
    assert np.array_equal(back["proof"], fx["proof"]) and np.array_equal(back["z"], fx["z"])
    for m in ("a", "b", "c"):
        for x, y in zip(back["r1cs"][m], fx["r1cs"][m]):
            assert np.array_equal(np.asarray(x, dtype=np.uint64), np.asarray(y, dtype=np.uint64))
    proof, inf = oracle.prove(back["pk"], back["r"], back["s"], back["r1cs"], back["z"])
    assert np.array_equal(proof, back["proof"]) and list(inf) == list(back["inf"])


@pytest.mark.skipif(not FIXTURES, reason="no arkworks-generated fixture present (tests/golden/ark_fixture_*.bin)")
@pytest.mark.parametrize("path", FIXTURES)
def test_oracle_reproduces_arkworks_proof(path, oracle):
    fx = ark_fixture.load(path)
    proof, inf = oracle.prove(fx["pk"], fx["r"], fx["s"], fx["r1cs"], fx["z"])
    assert np.array_equal(proof, fx["proof"]) and list(inf) == list(fx["inf"])


@pytest.mark.gpu
def test_device_reproduces_fixture_proof(tmp_path):
    """The HIP path on every fixture present — at least the self-written one."""
    from zksnark_finalproject_amd import Device
    dev = Device(0)
    paths = FIXTURES + [_own_fixture(str(tmp_path))[0]]
    for path in paths:
        fx = ark_fixture.load(path)
        ph = dev.pk_load(fx["pk"], fx["num_instance"])
        proof, inf = dev.prove(ph, fx["r"], fx["s"], fx["r1cs"], fx["z"])
        dev.pk_free(ph)
        assert np.array_equal(proof, fx["proof"]) and list(inf) == list(fx["inf"]), path
    dev.close()
