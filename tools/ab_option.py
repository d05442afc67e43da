"""A/B a library option in one process (development probe): python tools/ab_option.py <option> <v0,v1,...> [matrix_n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
opt = sys.argv[1]
vals = [int(x) for x in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = Device(0)
c = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(num_vars=c.num_vars, num_instance=c.num_instance, domain=c.domain)
pk = bench.make_key(dev, c.r1cs, shp, seed=1)
ph, rh, wh = dev.pk_load(pk, 4), dev.r1cs_load(c.r1cs, c.num_vars), dev.witness_load(c.z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
ref = dev.prove_resident(ph, rh, wh, r, s)
res = {v: [] for v in vals}
same = True
for rnd in range(6):
    for v in vals:
        dev.set_option(opt, v)
        out = dev.prove_resident(ph, rh, wh, r, s)
        same = same and np.array_equal(out[0], ref[0])
        t0 = time.perf_counter()
        for _ in range(5):
            dev.prove_resident(ph, rh, wh, r, s)
        res[v].append((time.perf_counter() - t0) / 5 * 1e3)
for v in vals:
    x = sorted(res[v])
    print("n=%d %s=%d  ms/proof min %.2f median %.2f max %.2f  (proofs identical: %s)" % (n, opt, v, x[0], x[len(x) // 2], x[-1], same))
