"""A/B library options in one process (development probe, GPU box):
   python tools/ab_option.py <matrix_n> [tables=CZ,CH] <opt=v0,v1,...> [<opt2=...> ...]     (options are swept one at a time, the others
   at 0/default; tables=: window tables first, 0 = default widths)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.device import verify
n = int(sys.argv[1])
sweeps = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in sys.argv[2:]]
tables = [v for k, v in sweeps if k == "tables"]
sweeps = [(k, v) for k, v in sweeps if k != "tables"]
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
ref = dev.prove_resident(ph, rh, wh, r, s)
print(desc, "verified:", verify(vk, c.public_inputs, *ref), flush=True)
if tables:
    print("window tables: %.2f GB" % (dev.pk_precompute(ph, *tables[0]) / 1e9), flush=True)
reps = 3 if n >= 100 else 5
for opt, vals in sweeps:
    res = {v: [] for v in vals}
    same = True
    for rnd in range(3 if n >= 100 else 6):
        for v in vals:
            dev.set_option(opt, v)
            out = dev.prove_resident(ph, rh, wh, r, s)
            same = same and np.array_equal(out[0], ref[0])
            t0 = time.perf_counter()
            for _ in range(reps):
                dev.prove_resident(ph, rh, wh, r, s)
            res[v].append((time.perf_counter() - t0) / reps * 1e3)
    dev.set_option(opt, 0 if opt not in ("wm_concurrent", "wm_first") else -1)
    for v in vals:
        x = sorted(res[v])
        print("n=%d %s=%d  ms/proof min %.2f median %.2f max %.2f  (proofs identical: %s)" % (n, opt, v, x[0], x[len(x) // 2], x[-1], same), flush=True)
