"""zkg16_prove (host pointers every call: the drop-in entry) vs zkg16_prove_resident (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(num_vars=c.num_vars, num_instance=c.num_instance, domain=c.domain)
dev = Device(0)
pk = bench.make_key(dev, c.r1cs, shp, seed=1)
ph = dev.pk_load(pk, c.num_instance)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
for _ in range(2):
    p1 = dev.prove(ph, r, s, c.r1cs, c.z)
t0 = time.perf_counter()
for _ in range(10):
    p1 = dev.prove(ph, r, s, c.r1cs, c.z)
t_host = (time.perf_counter() - t0) / 10
rh, wh = dev.r1cs_load(c.r1cs, c.num_vars), dev.witness_load(c.z)
for _ in range(2):
    p2 = dev.prove_resident(ph, rh, wh, r, s)
t0 = time.perf_counter()
for _ in range(10):
    p2 = dev.prove_resident(ph, rh, wh, r, s)
t_res = (time.perf_counter() - t0) / 10
print("n=%d zkg16_prove (host pointers) %.2f ms | zkg16_prove_resident %.2f ms | same proof %s" % (n, t_host * 1e3, t_res * 1e3, np.array_equal(p1[0], p2[0])))
