"""Second soak (development probe): many prime requests (threaded synthesis, circuit_load staging, threaded collects), then three
threads of mixed requests on one Device; every proof verified; free device memory watched."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from zksnark_finalproject_amd import Device, handlers
dev = Device(0)
free = lambda: torch.cuda.mem_get_info()[0] / 2**30
print("free at start %.1f GiB" % free(), flush=True)
bad = 0
t0 = time.perf_counter()
for it in range(150):
    x = 1000 + 7919 * it
    res = handlers.prove_prime(dev, x, 32)
    if not res["found_prime"]:
        continue
    ok = handlers.verify_prime(res["pvk"], x, res["j"], res["proof"])["valid"]
    wrong = handlers.verify_prime(res["pvk"], x + 1, res["j"], res["proof"])["valid"]
    bad += 0 if (ok and not wrong) else 1
    if it % 50 == 49:
        print("prime request %d, %.1f s, bad %d, free %.1f GiB" % (it + 1, time.perf_counter() - t0, bad, free()), flush=True)
errs = []
def worker(k):
    try:
        rng = np.random.default_rng(k)
        for it in range(25):
            n = [4, 8, 16, 32][(it + k) % 4]
            a = rng.integers(0, 1 << 20, size=(n, n), dtype=np.uint64)
            res = handlers.prove_matrix(dev, n, a, a, seed=it * 3 + k)
            if not handlers.verify_proof(res["vk"], res["_circuit"].public_inputs, res["proof"])["valid"]:
                errs.append(("invalid", k, it))
            if it % 5 == 0:
                r2 = handlers.prove_fibonacci(dev, 0, 1, 20 + it)
                if not handlers.verify_proof(r2["vk"], r2["_circuit"].public_inputs, r2["proof"])["valid"]:
                    errs.append(("invalid fib", k, it))
    except Exception as e:      # noqa: BLE001
        errs.append((repr(e), k))
for n in (4, 8, 16, 32):
    handlers.prove_matrix(dev, n, np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))      # shapes created before the threads share them
ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads done: errors %s; bad prime %d; free %.1f GiB; total %.1f s" % (errs, bad, free(), time.perf_counter() - t0), flush=True)
dev.close()
print("after close: free %.1f GiB" % free())
sys.exit(1 if (errs or bad) else 0)
