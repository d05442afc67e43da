"""Development timing probe (GPU box): NTT and MSM stage timings at the BASELINE sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np

from helpers import G1_GEN_LIMBS, G2_GEN_LIMBS
from zksnark_finalproject_amd import Device

dev = Device(0)
for log_n in (16, 19, 20):
    for inv, coset in ((0, 0), (1, 1)):
        print("ntt 2^%d inv=%d coset=%d: %.3f ms" % (log_n, inv, coset, dev.bench_ntt(log_n, inv, coset, 10)), flush=True)
rng = np.random.default_rng(1)
sizes = [int(x) for x in (sys.argv[1:] or ["65536", "524288"])]
for group, gen in (("g1", G1_GEN_LIMBS), ("g2", G2_GEN_LIMBS)):
    for n in sizes:
        logs = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        t0 = time.time()
        pts, inf = dev.fixed_base(group, gen, logs)
        t_fb = time.time() - t0
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        for c in (0, 13, 14, 15, 16):
            dev.set_option("window_bits", c)
            dev.kernel_stats_reset()
            dev.kernel_timing(True)
            ms, out, oinf = dev.bench_msm(group, pts, sc, iters=2, inf=inf)
            dev.kernel_timing(False)
            acc = dev.kernel_stats("msm_accumulate_" + group)
            red = dev.kernel_stats("msm_reduce_level_" + group)
            fix = dev.kernel_stats("msm_fixup_" + group)
            srt = sum(dev.kernel_stats(k)["ms"] for k in ("msm_digits_kernel", "msm_scan_kernel", "msm_scatter_kernel"))
            print("msm %s n=%d c=%d: %.2f ms/iter (timed-serial) | accumulate %.2f ms (%d launches, %.0f adds) reduce %.2f fixup %.2f sort %.2f | fixed_base %.2fs"
                  % (group, n, c, ms, acc["ms"] / max(acc["launches"], 1), acc["launches"], acc["units"] / max(acc["launches"], 1),
                     red["ms"] / 2, fix["ms"] / 2, srt / 2, t_fb), flush=True)
        dev.set_option("window_bits", 0)
        ms, _, _ = dev.bench_msm(group, pts, sc, iters=3, inf=inf)
        print("msm %s n=%d default c, no per-kernel timers: %.2f ms/iter" % (group, n, ms), flush=True)
