"""NTT: every stage through the LDS (option ntt_radix = 2) against the lane-exchange form of the last seven stages (1, the default)
and the radix-4 form (4): equality of the transforms, then time per transform (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device
dev = Device(0)
rng = np.random.default_rng(5)
for log_n in (4, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24):
    n = 1 << log_n
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    outs = {}
    for mode in (2, 1, 3):
        dev.set_option("ntt_radix", mode)
        outs[mode] = [dev.ntt(a, inv, coset) for inv, coset in ((False, True), (True, True), (True, False), (False, False))]
    same = all(np.array_equal(x, y) for x, y in zip(outs[2], outs[1])) and all(np.array_equal(x, y) for x, y in zip(outs[2], outs[3]))
    print("2^%d: lane-exchange transforms (modes 1 and 3) equal the all-LDS ones: %s" % (log_n, same), flush=True)
    if log_n < 19:
        continue
    for rep in range(2):
        for mode in (2, 1, 3, 4):
            dev.set_option("ntt_radix", mode)
            dev.bench_ntt(log_n, 1, 1, 2)
            ms = dev.bench_ntt(log_n, 1, 1, 20)
            print("2^%d ntt_radix=%d: %.3f ms per transform" % (log_n, mode, ms), flush=True)
dev.set_option("ntt_radix", 0)
