"""Repeated whole requests at one size (profiling target): python tools/setup_loop.py [matrix_n] [requests] [fixed_base_bits]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = Device(0)
if len(sys.argv) > 3:
    dev.set_option("fixed_base_bits", int(sys.argv[3]))
ones = np.ones((n, n), dtype=np.uint64)
for it in range(k):
    t0 = time.perf_counter()
    res = handlers.prove_matrix(dev, n, ones, ones, seed=it)
    print("request %.4f setup %.4f prove %.4f" % (time.perf_counter() - t0, res["setup_time"], res["proving_time"]), flush=True)
