"""Standalone NTT timings by mode (development probe): python tools/ntt_timing.py [log sizes...]
   mode 0 = saturated limbs, 1 = unsaturated (two 4096-point passes at 2^23-2^24), 3 = unsaturated with three passes there"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zksnark_finalproject_amd import Device
dev = Device(0)
for log_n in [int(x) for x in (sys.argv[1:] or ["16", "19", "20", "22", "23", "24"])]:
    for mode in (0, 3, 1):
        dev.set_option("ntt_mode", mode)
        dev.bench_ntt(log_n, 1, 1, 2)
        ms = dev.bench_ntt(log_n, 1, 1, 10)
        n = 1 << log_n
        print("ntt 2^%d coset-inverse ntt_mode=%d: %.3f ms  (%.1f GB/s of the 64 B/element algorithmic traffic)" % (log_n, mode, ms, 64.0 * n / ms / 1e6), flush=True)
