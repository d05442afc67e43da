"""Does a small plain-key proof slow down after a large workload ran on the same ctx? (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
def setup(n):
    c, _, _ = bench.synthesize("matrix", n)
    rh = dev.r1cs_load(c.r1cs, c.num_vars)
    ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
    wh = dev.witness_load(c.z)
    return ph, rh, wh
def timed(tag, h, reps=10):
    dev.prove_resident(*h, r, s)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.prove_resident(*h, r, s)
    print("%-40s %.2f ms/proof  %s" % (tag, (time.perf_counter() - t0) / reps * 1e3, {k: round(v, 2) for k, v in dev.last_timings().items() if isinstance(v, float)}), flush=True)
small = setup(32)
timed("32x32 plain, fresh ctx", small)
big = setup(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
timed("big plain", big, 2)
timed("32x32 plain, after the big one", small)
dev.pk_precompute(big[0])
timed("big tabled", big, 2)
timed("32x32 plain, after the big tabled one", small)
for h in big:
    pass
dev.pk_free(big[0]); dev.witness_free(big[2]); dev.r1cs_free(big[1])
timed("32x32 plain, big one freed", small)
