"""One MSM alone on the device (no other stream active): time of the accumulation kernel against its exact term count
(development probe).   python tools/msm_alone.py [log2 n] [g1|g2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device
from zksnark_finalproject_amd.workloads import g1_generator, g2_generator
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
group = sys.argv[2] if len(sys.argv) > 2 else "g1"
n = 1 << log_n
dev = Device(0)
rng = np.random.default_rng(5)
logs = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
pts, inf = dev.fixed_base(group, g1_generator() if group == "g1" else g2_generator(), logs)
sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64(0x3fffffffffffffff)
dev.bench_msm(group, pts, sc, 1, inf)
for waves in ((2, 4) if group == "g1" else (0,)):
    if waves:
        dev.set_option("g1_waves", waves)
    dev.kernel_stats_reset()
    dev.kernel_timing(True)
    ms = dev.bench_msm(group, pts, sc, 3, inf)[0]
    dev.kernel_timing(False)
    k = "msm_accumulate_" + group
    st = dev.kernel_stats(k)
    print("%s n=2^%d waves/SIMD=%s: MSM %.2f ms; accumulation kernel %.3f ms per launch" % (group, log_n, waves or 1, ms, st["ms"] / max(st["launches"], 1)), flush=True)
    for name in ("msm_digits_kernel", "msm_bucket_sort", "msm_offsets_kernel", "msm_fixup_" + group, "msm_fixup_long_" + group, "msm_reduce_" + group):
        s2 = dev.kernel_stats(name)
        if s2["launches"]:
            print("     %-22s %.3f ms" % (name, s2["ms"] / s2["launches"]))
