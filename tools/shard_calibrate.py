"""Calibration data for zkg16_shard_plan's cost model (development probe, one GPU): time of zkg16_prove_partial for z-only shards
holding a given fraction of the z-side work (by zkg16's per-variable cost) and for h-only shards holding a fraction of h_query.
   python tools/shard_calibrate.py [matrix_n] [tables]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.device import z_costs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tables = len(sys.argv) > 2 and sys.argv[2] == "tables"
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
full, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
reps = 3 if n >= 100 else 8
cum = np.cumsum(z_costs(c.r1cs, c.z, c.num_instance).astype(np.float64))
m, nh = c.num_vars, c.domain - 1
print(desc + ("  [window tables]" if tables else ""), " total z cost %.4g G1 additions, h terms %d" % (cum[-1], nh), flush=True)
def timed(z_lo, z_hi, h_lo, h_hi, blind):
    sh = dev.pk_slice(full, z_lo, z_hi, h_lo, h_hi, blind)
    if tables:
        dev.pk_precompute(sh)
    dev.prove_partial(sh, rh, wh, r, s)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.prove_partial(sh, rh, wh, r, s)
    dt = (time.perf_counter() - t0) / reps * 1e3
    dev.pk_free(sh)
    return dt
for f in (1 / 32, 1 / 16, 1 / 8, 1 / 4, 1 / 2):
    lo = int(np.searchsorted(cum, 0.25 * cum[-1]))
    hi = int(np.searchsorted(cum, (0.25 + f) * cum[-1]))
    print("z-only shard, %.4f of the z cost (%d variables): %.2f ms" % (f, hi - lo, timed(lo, hi, 0, 0, True)), flush=True)
for g in (1 / 8, 1 / 5, 1 / 4, 1 / 2, 1.0):
    print("h-only shard, %.3f of h_query: %.2f ms" % (g, timed(0, 0, 0, int(nh * g), False)), flush=True)
