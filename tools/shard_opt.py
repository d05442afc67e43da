"""Per-rank partial-proof time for shard 0 of G under an option sweep (development probe):
   python tools/shard_opt.py <option> <v0,v1,..> [matrix_n] [G]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.circuits import matrix_circuit
opt = sys.argv[1]; vals = [int(x) for x in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
G = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = Device(0)
c = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
shp = dict(num_vars=c.num_vars, num_instance=c.num_instance, domain=c.domain)
pk = bench.make_key(dev, c.r1cs, shp, seed=1)
rh, wh = dev.r1cs_load(c.r1cs, c.num_vars), dev.witness_load(c.z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
ph = dev.pk_load(pk, c.num_instance, shard_index=0, shard_count=G)
ref = dev.prove_partial(ph, rh, wh, r, s)
for rnd in range(2):
    for v in vals:
        dev.set_option(opt, v)
        out = dev.prove_partial(ph, rh, wh, r, s)
        t0 = time.perf_counter()
        for _ in range(10):
            dev.prove_partial(ph, rh, wh, r, s)
        print("n=%d shard 0 of %d %s=%d: %.2f ms  same=%s" % (n, G, opt, v, (time.perf_counter() - t0) / 10 * 1e3, np.array_equal(out[0], ref[0])), flush=True)
