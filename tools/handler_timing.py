"""Request-level timing of the handler mirror (setup + prove per request, as the reference does): development probe.
   python tools/handler_timing.py [matrix_n]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers, wire
from zksnark_finalproject_amd.circuits import matrix_circuit

dev = Device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ones = np.ones((n, n), dtype=np.uint64)
for it in range(3):
    t0 = time.perf_counter()
    res = handlers.prove_matrix(dev, n, ones, ones, seed=it)
    t1 = time.perf_counter()
    v = handlers.verify_proof(res["_detail"]["vk"], res["_circuit"].public_inputs, res["proof"])
    print("n=%d request %.3f s: setup %.3f s, prove %.4f s, verify %.3f s valid=%s, synth+encode %.3f s" %
          (n, t1 - t0, res["setup_time"], res["proving_time"], v["verifying_time"], v["valid"], t1 - t0 - res["setup_time"] - res["proving_time"]), flush=True)
# where the host side of a request goes
t = [time.perf_counter()]
circ = matrix_circuit(ones, ones); t.append(time.perf_counter())
rh = dev.r1cs_load(circ.r1cs, circ.num_vars); t.append(time.perf_counter())
wh = dev.witness_load(circ.z); t.append(time.perf_counter())
enc = wire.encode_proof(res["_detail"]["proof"], res["_detail"]["inf"]); t.append(time.perf_counter())
print("host side: synthesize+export %.3f s, r1cs_load %.3f s, witness_load %.3f s, encode_proof %.4f s" %
      tuple(t[i + 1] - t[i] for i in range(4)))
