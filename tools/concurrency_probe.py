"""Throughput with C proofs in flight on one GPU (one ctx per host thread, as one ctx per actix worker would be):
   python tools/concurrency_probe.py [matrix_n] [proofs_per_thread]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
circ = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(n=n, nc=circ.num_constraints, num_instance=circ.num_instance, num_witness=circ.num_witness, num_vars=circ.num_vars, domain=circ.domain)
rng = np.random.default_rng(5)
rs = [(bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)) for _ in range(K + 2)]
for C in (1, 2, 3):
    devs = [Device(0) for _ in range(C)]
    hs = []
    for d in devs:
        pk = bench.make_key(d, circ.r1cs, shp, seed=0xC0FFEE)
        hs.append((d.pk_load(pk, shp["num_instance"]), d.r1cs_load(circ.r1cs, shp["num_vars"]), d.witness_load(circ.z)))
        del pk
    outs = [None] * C
    def work(i, count):
        d, (ph, rh, wh) = devs[i], hs[i]
        for j in range(count):
            outs[i] = d.prove_resident(ph, rh, wh, *rs[j])
    for i in range(C):
        work(i, 2)
    ths = [threading.Thread(target=work, args=(i, K)) for i in range(C)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    same = all(np.array_equal(outs[0][0], o[0]) for o in outs)
    print("n=%d contexts=%d: %d proofs in %.3f s = %.1f proofs/s (%.2f ms/proof), proofs identical across contexts: %s" % (n, C, C * K, dt, C * K / dt, dt / (C * K) * 1e3, same), flush=True)
    for d in devs: d.close()
