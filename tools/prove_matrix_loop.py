"""A bare loop of streamed matrix requests (profiling target): python tools/prove_matrix_loop.py [n] [requests] [tables on|off] [opt=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tables = (sys.argv[3] if len(sys.argv) > 3 else "on") == "on"
dev = Device(0)
for a in sys.argv[4:]:
    dev.set_option(a.split("=")[0], int(a.split("=")[1]))
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
if tables:
    dev.pk_precompute(ph)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
ones = np.ones((n, n), dtype=np.uint64)
for _ in range(k):
    t0 = time.perf_counter()
    p, i, pub, ms = dev.prove_matrix(ph, rh, ones, ones, r, s)
    print("request %.1f ms" % ((time.perf_counter() - t0) * 1e3), ms, flush=True)
print("done", desc)
