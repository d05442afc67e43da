"""Proof latency on the reference's small circuits (development probe)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device, handlers
from zksnark_finalproject_amd.circuits import fibonacci_circuit, matrix_circuit
dev = Device(0)
for name, circ in (("fibonacci 186 rounds", fibonacci_circuit(0, 1, 186)), ("fibonacci 1000 rounds", fibonacci_circuit(0, 1, 1000)),
                   ("matrix 4x4", matrix_circuit(np.ones((4, 4), dtype=np.uint64), np.ones((4, 4), dtype=np.uint64)))):
    shp = dict(num_vars=circ.num_vars, num_instance=circ.num_instance, domain=circ.domain)
    pk = bench.make_key(dev, circ.r1cs, shp, seed=3)
    ph, rh, wh = dev.pk_load(pk, circ.num_instance), dev.r1cs_load(circ.r1cs, circ.num_vars), dev.witness_load(circ.z)
    rng = np.random.default_rng(5)
    r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
    for _ in range(3):
        dev.prove_resident(ph, rh, wh, r, s)
    t0 = time.perf_counter()
    for _ in range(20):
        dev.prove_resident(ph, rh, wh, r, s)
    dt = (time.perf_counter() - t0) / 20
    print("%s: %d constraints, domain 2^%d: %.3f ms/proof; stages %s" % (name, circ.num_constraints, circ.domain.bit_length() - 1, dt * 1e3,
          {k: round(v, 3) for k, v in dev.last_timings().items()}), flush=True)
