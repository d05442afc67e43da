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
                   ("matrix 4x4", matrix_circuit(np.ones((4, 4), dtype=np.uint64), np.ones((4, 4), dtype=np.uint64))),
                   ("prime circuit", __import__("zksnark_finalproject_amd.circuits", fromlist=["prime_circuit"]).prime_circuit(5, 32))):
    trap, g1, g2 = bench.draw_key_inputs(3)
    rh, wh = dev.r1cs_load(circ.r1cs, circ.num_vars), dev.witness_load(circ.z)
    ph, vk = dev.setup_resident(rh, circ.num_instance, trap, g1, g2)
    r, s = bench.fr_mont(12345), bench.fr_mont(67890)
    for _ in range(3):
        dev.prove_resident(ph, rh, wh, r, s)
    t0 = time.perf_counter()
    for _ in range(20):
        dev.prove_resident(ph, rh, wh, r, s)
    dt = (time.perf_counter() - t0) / 20
    print("%s: %d constraints, domain 2^%d: %.3f ms/proof; stages %s" % (name, circ.num_constraints, circ.domain.bit_length() - 1, dt * 1e3,
          {k: round(v, 3) for k, v in dev.last_timings().items() if isinstance(v, float)}), flush=True)
