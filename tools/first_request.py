"""First-request cost at a large size after smaller ones in the same process (development probe): python tools/first_request.py 16,64,128,128"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers
dev = Device(0)
if len(sys.argv) > 2:
    dev.set_option("fixed_base_bits", int(sys.argv[2]))
for n in [int(x) for x in sys.argv[1].split(",")]:
    ones = np.ones((n, n), dtype=np.uint64)
    t0 = time.perf_counter()
    res = handlers.prove_matrix(dev, n, ones, ones, seed=n)
    print("n=%d request %.4f setup %.4f prove %.4f" % (n, time.perf_counter() - t0, res["setup_time"], res["proving_time"]), flush=True)
