"""A/B: window bits of the H MSM's own plan (development probe): python tools/ab_hwin.py [matrix_n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
circ = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(n=n, nc=circ.num_constraints, num_instance=circ.num_instance, num_witness=circ.num_witness, num_vars=circ.num_vars, domain=circ.domain)
dev = Device(0)
pk = bench.make_key(dev, circ.r1cs, shp, seed=0xC0FFEE)
ph, rh, wh = dev.pk_load(pk, shp["num_instance"]), dev.r1cs_load(circ.r1cs, shp["num_vars"]), dev.witness_load(circ.z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
ref = None
for rep in range(2):
    for c in (0, 10, 11, 12, 13, 14):
        dev.set_option("window_bits_h", c)
        out = dev.prove_resident(ph, rh, wh, r, s)
        t0 = time.perf_counter()
        for _ in range(10):
            out = dev.prove_resident(ph, rh, wh, r, s)
        dt = (time.perf_counter() - t0) / 10
        if ref is None: ref = out
        print("n=%d window_bits_h=%d: %.2f ms/proof same=%s" % (n, c, dt * 1e3, np.array_equal(out[0], ref[0])), flush=True)
