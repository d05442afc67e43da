"""Small matrix requests: where the time goes (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers
dev = Device(0)
for n in (2, 4, 8, 16, 32, 8, 8):
    ones = np.ones((n, n), dtype=np.uint64)
    handlers.prove_matrix(dev, n, ones, ones, seed=1)
    t0 = time.perf_counter(); res = handlers.prove_matrix(dev, n, ones, ones, seed=2); t1 = time.perf_counter()
    w = []
    for _ in range(3):
        a = time.perf_counter(); wh, pub, ms = dev.witness_matrix(ones, ones); b = time.perf_counter(); dev.witness_free(wh)
        w.append((b - a) * 1e3)
    print("n=%d request %.2f ms setup %.2f prove %.2f | witness_matrix alone %.2f ms %s" % (n, (t1 - t0) * 1e3, res["setup_time"] * 1e3, res["proving_time"] * 1e3, min(w), {k: round(v, 2) for k, v in ms.items()}), flush=True)
