"""A bare loop of partial proofs for shard k of G (profiling target): python tools/prove_loop_shard.py [matrix_n] [G] [k] [proofs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 5
circ = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(n=n, nc=circ.num_constraints, num_instance=circ.num_instance, num_witness=circ.num_witness, num_vars=circ.num_vars, domain=circ.domain)
dev = Device(0)
pk = bench.make_key(dev, circ.r1cs, shp, seed=0xC0FFEE)
ph, rh, wh = dev.pk_load(pk, shp["num_instance"], shard_index=k, shard_count=G), dev.r1cs_load(circ.r1cs, shp["num_vars"]), dev.witness_load(circ.z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
for _ in range(cnt):
    dev.prove_partial(ph, rh, wh, r, s)
print("done", dev.last_timings())
