"""Per-kernel HIP-event table for one workload (development probe, GPU box):
   python tools/kstats.py [matrix_n] [window_bits] [reduce_chunk]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.workloads import matmul_like_r1cs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = Device(0)
if len(sys.argv) > 2:
    dev.set_option("window_bits", int(sys.argv[2]))
if len(sys.argv) > 3:
    dev.set_option("reduce_chunk", int(sys.argv[3]))
r1cs, z, shp = matmul_like_r1cs(n)
pk = bench.make_key(dev, r1cs, shp, seed=1)
ph, rh, wh = dev.pk_load(pk, 4), dev.r1cs_load(r1cs, shp["num_vars"]), dev.witness_load(z)
rng = np.random.default_rng(5)
r, s = bench.rand_fr_mont(rng), bench.rand_fr_mont(rng)
dev.prove_resident(ph, rh, wh, r, s)
dev.kernel_stats_reset()
dev.kernel_timing(True)
reps = 2
for _ in range(reps):
    dev.prove_resident(ph, rh, wh, r, s)
dev.kernel_timing(False)
names = ["spmv_kernel", "ntt_pass_cols", "ntt_pass_rows", "pointwise_h_kernel", "fr_from_mont_kernel", "msm_digits_kernel", "msm_radix_sort", "msm_offsets_kernel",
         "msm_accumulate_g1", "msm_fixup_g1", "msm_fixup_long_g1", "msm_reduce_g1", "msm_accumulate_g2", "msm_fixup_g2",
         "msm_fixup_long_g2", "msm_reduce_g2"]
tot = 0
for k in names:
    st = dev.kernel_stats(k)
    if st["launches"]:
        print("%-22s launches/proof %5.1f  ms/proof %8.3f  avg_ms %8.4f" % (k, st["launches"] / reps, st["ms"] / reps, st["ms"] / st["launches"]))
        tot += st["ms"] / reps
print("sum of kernels: %.2f ms/proof;  stages:" % tot, dev.last_timings())
