"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code only; GPU sanitizers are not available on the
pool).  oracle/Makefile builds libg16oracle_asan.so; a child interpreter preloads the sanitizer runtimes, runs NTTs, MSMs, a
witness map and one whole proof on it, and must finish with the same proof as the optimised build and no sanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys, random
sys.path[:0] = [ROOT, ROOT + "/tests", ROOT + "/tests/golden", ROOT + "/oracle"]
import numpy as np
import oracle, synth, pyref as P
from helpers import fr_mont, fr_mont_vec
rng = random.Random(11)
nc, ni, nv = 40, 3, 30
A, B, C, z = synth.random_r1cs(rng, nc, ni, nv)
r1cs = synth.r1cs_arrays(A, B, C, ni)
pk, _ = synth.make_pk(oracle, r1cs, nv, rng)
zm = fr_mont_vec(z)
x = fr_mont_vec([P.rand_fr(rng) for _ in range(64)])
for inv in (False, True):
    for coset in (False, True):
        oracle.ntt(x, inv, coset)
h = oracle.witness_map(r1cs, zm)
proof, inf = oracle.prove(pk, fr_mont(5), fr_mont(7), r1cs, zm)
np.save(OUT, np.concatenate([proof, inf.astype(np.uint64)]))
print("child ok")
'''


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("gcc has no libasan.so in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "build/libg16oracle_asan.so"])
    lib = os.path.join(ROOT, "oracle", "build", "libg16oracle_asan.so")
    outs = {}
    for tag, env_extra in (("asan", {"ZKG16_ORACLE_LIB": lib, "LD_PRELOAD": ":".join(x for x in (asan, ubsan) if x),
                                      "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}),
                           ("plain", {})):
        out = tmp_path / (tag + ".npy")
        script = tmp_path / (tag + ".py")
        script.write_text("ROOT = %r\nOUT = %r\n" % (ROOT, str(out)) + _CHILD)
        env = dict(os.environ, OMP_NUM_THREADS="2", **env_extra)
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
        import numpy as np
        outs[tag] = np.load(out)
    assert (outs["asan"] == outs["plain"]).all()
