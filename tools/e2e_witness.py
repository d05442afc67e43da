#!/usr/bin/env python3
"""Where the end-to-end (cached matrices) request of the 128x128 MatrixCircuit spends its time: host sponges, device
assignment kernels, resident proof — against the host-built assignment + upload it replaces.  python tools/e2e_witness.py [n] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import draw_key_inputs, fr_mont  # noqa: E402
from zksnark_finalproject_amd import Device  # noqa: E402
from zksnark_finalproject_amd.circuits import matrix_circuit, matrix_sponge_states, matrix_witness  # noqa: E402
from zksnark_finalproject_amd.device import verify  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tables = (sys.argv[3] if len(sys.argv) > 3 else "on") == "on"
ones = np.ones((n, n), dtype=np.uint64)
print("host cores:", os.cpu_count())
for _ in range(reps):
    t = time.perf_counter()
    matrix_sponge_states(ones, ones)
    print("host sponges alone (3 threads): %.1f ms" % ((time.perf_counter() - t) * 1e3))
dev = Device(0)
trap, g1, g2 = draw_key_inputs(2026)
circ = matrix_circuit(ones, ones)
rh = dev.r1cs_load(circ.r1cs, circ.num_vars)
ph, vk = dev.setup_resident(rh, circ.num_instance, trap, g1, g2)
if tables:
    dev.pk_precompute(ph)
r, s = fr_mont(12345), fr_mont(67890)
wh0 = dev.witness_load(circ.z)
dev.prove_resident(ph, rh, wh0, r, s)
for _ in range(reps):
    t0 = time.perf_counter()
    z = matrix_witness(ones, ones, circ.num_vars)
    t1 = time.perf_counter()
    w = dev.witness_load(z)
    t2 = time.perf_counter()
    p_old = dev.prove_resident(ph, rh, w, r, s)
    t3 = time.perf_counter()
    dev.witness_free(w)
    print("host assignment %.1f + upload %.1f + prove %.1f = %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
for _ in range(reps):
    t0 = time.perf_counter()
    w, pub, ms = dev.witness_matrix(ones, ones)
    t1 = time.perf_counter()
    p_new = dev.prove_resident(ph, rh, w, r, s)
    t2 = time.perf_counter()
    dev.witness_free(w)
    print("device assignment %.1f (sponges %.1f, device %.2f) + prove %.1f = %.1f ms" %
          ((t1 - t0) * 1e3, ms["host_sponges_ms"], ms["device_ms"], (t2 - t1) * 1e3, (t2 - t0) * 1e3))
for parts in (0, 2, 3, 6, 8, 1):
    dev.set_option("matrix_parts", parts)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        p_st, i_st, pub, ms = dev.prove_matrix(ph, rh, ones, ones, r, s)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    print("prove_matrix matrix_parts=%d (parts used %d): best %.1f ms (sponges %.1f ms overlapped), same proof: %s" %
          (parts, ms["parts"], best, ms["host_sponges_ms"], bool(np.array_equal(p_st, p_new[0]))))
dev.set_option("matrix_parts", 0)
print("same proof:", bool(np.array_equal(p_old[0], p_new[0])), "verified:", verify(vk, circ.public_inputs, *p_new))
dev.close()
