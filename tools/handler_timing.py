"""Request-level timing of the handler mirror (setup + prove per request, as the reference does): development probe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers
dev = Device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for it in range(3):
    t0 = time.perf_counter()
    res = handlers.prove_matrix(dev, n, np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64), seed=it)
    t1 = time.perf_counter()
    v = handlers.verify_proof(res["_detail"]["vk"], res["_circuit"].public_inputs, res["proof"])
    print("n=%d request %.3f s: setup %.3f s, prove %.4f s, verify %.3f s valid=%s, synth+encode %.3f s" %
          (n, t1 - t0, res["setup_time"], res["proving_time"], v["verifying_time"], v["valid"], t1 - t0 - res["setup_time"] - res["proving_time"]), flush=True)
