"""A bare loop of resident proofs (profiling target): python tools/prove_loop.py [matrix_n] [proofs] [opt=value ...]
   (tables=CZ,CH: window tables first, 0 = default widths)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = Device(0)
tables = None
for a in sys.argv[3:]:
    if a.startswith("tables="):
        tables = [int(x) for x in a.split("=")[1].split(",")]
    else:
        dev.set_option(a.split("=")[0], int(a.split("=")[1]))
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
if tables:
    dev.pk_precompute(ph, *tables)
for _ in range(k):
    dev.prove_resident(ph, rh, wh, r, s)
print("done", desc, dev.last_timings())
