"""Average launch time of the two bucket-accumulation kernels under a library option (development probe):
   python tools/acc_probe.py <matrix_n> <opt=v0,v1,...> [...]      e.g.  acc_debug=0,1,2,3  (1: no bucket stores, 2: no gather; WRONG results)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device
n = int(sys.argv[1])
sweeps = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in sys.argv[2:]]
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
dev.prove_resident(ph, rh, wh, r, s)
print(desc, flush=True)
reps = 3 if n >= 100 else 6
for opt, vals in sweeps:
    for v in vals:
        dev.set_option(opt, v)
        dev.prove_resident(ph, rh, wh, r, s)
        dev.kernel_stats_reset()
        dev.kernel_timing(2)
        for _ in range(reps):
            dev.prove_resident(ph, rh, wh, r, s)
        dev.kernel_timing(False)
        a1, a2 = dev.kernel_stats("msm_accumulate_g1"), dev.kernel_stats("msm_accumulate_g2")
        print("n=%d %s=%d: accumulate g1 avg %.3f ms x %d per proof, g2 avg %.3f ms; proof wall %.2f ms" %
              (n, opt, v, a1["ms"] / max(a1["launches"], 1), a1["launches"] // reps, a2["ms"] / max(a2["launches"], 1), dev.last_timings()["total_wall"]), flush=True)
    dev.set_option(opt, 0)
