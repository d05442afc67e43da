"""Per-kernel HIP-event table for one workload (development probe, GPU box):
   python tools/kstats.py [matrix_n] [opt=value ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = Device(0)
for a in sys.argv[2:]:
    dev.set_option(a.split("=")[0], int(a.split("=")[1]))
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix", n)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
dev.prove_resident(ph, rh, wh, r, s)
dev.kernel_stats_reset()
dev.kernel_timing(True)
reps = 2
for _ in range(reps):
    dev.prove_resident(ph, rh, wh, r, s)
dev.kernel_timing(False)
names = ["spmv_kernel", "ntt_pass_cols", "ntt_pass_rows", "pointwise_h_kernel", "msm_digits_kernel", "msm_bucket_sort", "msm_radix_sort", "msm_offsets_kernel",
         "msm_accumulate_g1", "msm_fixup_g1", "msm_fixup_long_g1", "msm_reduce_g1", "msm_accumulate_g2", "msm_fixup_g2",
         "msm_fixup_long_g2", "msm_reduce_g2"]
tot = 0
print(desc)
for k in names:
    st = dev.kernel_stats(k)
    if st["launches"]:
        print("%-22s launches/proof %5.1f  ms/proof %8.3f  avg_ms %8.4f" % (k, st["launches"] / reps, st["ms"] / reps, st["ms"] / st["launches"]))
        tot += st["ms"] / reps
print("sum of kernels (they overlap: each figure includes waiting for the others): %.2f ms/proof;  stages:" % tot, dev.last_timings())
