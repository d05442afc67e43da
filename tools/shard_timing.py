"""Per-rank time of one sharded proof under zkg16_shard_plan's rank roles, measured on ONE GPU that plays every rank in turn
(development probe; no collective involved: the exchange is one 77-word all_gather, ~0.1 ms):
   python tools/shard_timing.py [matrix_n] [G,G,...] [tables]      (tables: every shard gets window tables, zkg16_pk_precompute)
For each G: the plan the cost model picks (k ranks run the witness map), the plan with k forced to G (round 1's equal split),
and per rank the time of zkg16_prove_partial.  The slowest rank is the proof's time; speed-up = single-GPU time / that."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.device import shard_plan, z_costs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Gs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
tables = len(sys.argv) > 3 and sys.argv[3] == "tables"
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
full, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
reps = 3 if n >= 100 else 8
costs = z_costs(c.r1cs, c.z, c.num_instance)
print(desc + ("  [window tables per shard]" if tables else ""), flush=True)
single = None
if 1 not in Gs:
    Gs = [1] + Gs
ks = [int(x) for a in sys.argv[4:] if a.startswith("k=") for x in a[2:].split(",")]      # k=4,5: also time these witness-map rank counts (cost-based z cuts)
for G in Gs:
    for force in ([0] if G == 1 else [0, G] + [-kk for kk in ks if kk < G]):
        by_cost = force <= 0
        plan, k = shard_plan(G, c.num_vars, c.domain - 1, 0.0, abs(force), costs if by_cost else None, window_tables=tables)
        worst = 0.0
        per_rank = []
        seen = {}
        for i, (z_lo, z_hi, h_lo, h_hi, blind) in enumerate(plan):
            sig = (z_hi - z_lo, h_hi - h_lo, blind)
            if sig in seen and G > 2:          # ranks with the same amount of work: measure one of them
                per_rank.append(seen[sig])
                continue
            sh = dev.pk_slice(full, z_lo, z_hi, h_lo, h_hi, blind)
            if tables:
                dev.pk_precompute(sh)
            dev.prove_partial(sh, rh, wh, r, s)
            t0 = time.perf_counter()
            for _ in range(reps):
                dev.prove_partial(sh, rh, wh, r, s)
            dt = (time.perf_counter() - t0) / reps * 1e3
            dev.pk_free(sh)
            seen[sig] = dt
            per_rank.append(dt)
        worst = max(per_rank)
        if G == 1:
            single = worst
        print("n=%d G=%d witness-map ranks k=%d%s: per-rank ms %s -> slowest %.2f ms, speed-up %.2fx" %
              (n, G, k, " (forced: equal split)" if force > 0 else " (forced k, cost-based z cuts)" if force < 0 else " (cost model)", [round(x, 2) for x in per_rank], worst, single / worst), flush=True)
