"""Window tables A/B in one process (development probe, GPU box): proofs with the plain resident key, then the same key after
zkg16_pk_precompute.    python tools/ab_tables.py <matrix_n> [window_bits_z [window_bits_h]] [opt=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.device import verify
args = [a for a in sys.argv[1:] if "=" not in a]
n = int(args[0])
cz = int(args[1]) if len(args) > 1 else 0
ch = int(args[2]) if len(args) > 2 else 0
dev = Device(0)
for a in sys.argv[1:]:
    if "=" in a:
        dev.set_option(a.split("=")[0], int(a.split("=")[1]))
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
ref = dev.prove_resident(ph, rh, wh, r, s)
print(desc, "verified:", verify(vk, c.public_inputs, *ref), flush=True)
reps = 3 if n >= 100 else 8
def timed(tag):
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            p = dev.prove_resident(ph, rh, wh, r, s)
        out.append((time.perf_counter() - t0) / reps * 1e3)
    print("%-28s ms/proof min %.2f median %.2f   same proof: %s   stages %s" % (tag, min(out), sorted(out)[1], np.array_equal(p[0], ref[0]),
          {k: round(v, 2) for k, v in dev.last_timings().items() if isinstance(v, float)}), flush=True)
timed("plain key")
t0 = time.perf_counter()
added = dev.pk_precompute(ph, cz, ch)
print("pk_precompute(%d, %d): %.3f s, %.2f GB of tables" % (cz, ch, time.perf_counter() - t0, added / 1e9), flush=True)
timed("with window tables")
