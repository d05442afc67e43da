"""Stability soak (development probe): repeated whole requests of varying sizes on one ctx, then proofs from three threads,
checking every proof with the pairing verifier and watching free device memory."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from zksnark_finalproject_amd import Device, handlers

def free_gb():
    return torch.cuda.mem_get_info()[0] / 2**30

dev = Device(0)
print("free at start %.1f GiB" % free_gb(), flush=True)
t0 = time.perf_counter()
bad = 0
for it in range(40):
    n = [3, 8, 16, 5, 32, 12, 24, 2][it % 8]
    rng = np.random.default_rng(it)
    a = rng.integers(0, 1 << 16, size=(n, n), dtype=np.uint64)
    b = rng.integers(0, 1 << 16, size=(n, n), dtype=np.uint64)
    res = handlers.prove_matrix(dev, n, a, b, seed=it)
    ok = handlers.verify_proof(res["vk"], res["_circuit"].public_inputs, res["proof"])["valid"]
    bad += 0 if ok else 1
    if it % 8 == 7:
        print("request %d done, %.1f s, invalid so far %d, free %.1f GiB" % (it + 1, time.perf_counter() - t0, bad, free_gb()), flush=True)
for rounds in (0, 7, 186):
    res = handlers.prove_fibonacci(dev, 0, 1, rounds)
    bad += 0 if handlers.verify_proof(res["vk"], res["_circuit"].public_inputs, res["proof"])["valid"] else 1
errs = []
def worker(k):
    try:
        d = Device(0)
        for it in range(12):
            n = [8, 16, 4][(it + k) % 3]
            ones = np.ones((n, n), dtype=np.uint64)
            res = handlers.prove_matrix(d, n, ones, ones, seed=100 * k + it)
            if not handlers.verify_proof(res["vk"], res["_circuit"].public_inputs, res["proof"])["valid"]:
                errs.append((k, it))
        d.close()
    except Exception as e:   # noqa
        errs.append((k, repr(e)))
ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
for t in ths: t.start()
for t in ths: t.join()
print("threads done: errors %s; invalid proofs %d; free %.1f GiB; total %.1f s" % (errs, bad, free_gb(), time.perf_counter() - t0), flush=True)
dev.close()
print("after close: free %.1f GiB" % free_gb())
assert not errs and bad == 0
