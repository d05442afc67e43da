"""A/B in one process: witness map in-order vs on a third stream (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = Device(0)
c = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(num_vars=c.num_vars, num_instance=c.num_instance, domain=c.domain)
pk = bench.make_key(dev, c.r1cs, shp, seed=1)
ph, rh, wh = dev.pk_load(pk, 4), dev.r1cs_load(c.r1cs, c.num_vars), dev.witness_load(c.z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
for _ in range(2):
    dev.prove_resident(ph, rh, wh, r, s)
res = {0: [], 1: []}
for rnd in range(6):
    for mode in (0, 1):
        dev.set_option("wm_concurrent", mode)
        t0 = time.perf_counter()
        for _ in range(5):
            dev.prove_resident(ph, rh, wh, r, s)
        res[mode].append((time.perf_counter() - t0) / 5 * 1e3)
for mode in (0, 1):
    v = sorted(res[mode])
    print("wm_concurrent=%d  ms/proof min %.2f median %.2f max %.2f" % (mode, v[0], v[len(v) // 2], v[-1]))
