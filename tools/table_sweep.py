"""Window-table widths swept on one circuit (development probe, GPU box): a fresh resident key per (cz, ch) pair.
   python tools/table_sweep.py <matrix_n|prime|fibN> cz:ch [cz:ch ...]      (-1 = no table on that side, 0 = default)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
kind = sys.argv[1]
pairs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]]
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix" if kind.isdigit() else kind, int(kind) if kind.isdigit() else 0)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
print(desc, flush=True)
ref = None
for cz, ch in pairs:
    ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
    if ref is None:
        ref = dev.prove_resident(ph, rh, wh, r, s)
    added = dev.pk_precompute(ph, cz, ch) if (cz, ch) != (-1, -1) else 0
    reps = 3 if c.num_constraints > 5e6 else 10
    out = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(reps):
            p = dev.prove_resident(ph, rh, wh, r, s)
        out.append((time.perf_counter() - t0) / reps * 1e3)
    print("tables z=%d h=%d -> %s, %.2f GB: ms/proof min %.2f median %.2f  same proof %s" % (cz, ch, dev.pk_table_bits(ph), added / 1e9, min(out), sorted(out)[1],
          np.array_equal(p[0], ref[0])), flush=True)
    dev.pk_free(ph)
