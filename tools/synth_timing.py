"""Host synthesis of the MatrixCircuit, phase by phase (CPU only; development probe): python tools/synth_timing.py [n]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import _lib, circuits
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ones = np.ones((n, n), dtype=np.uint64)
lib = _lib.load()
for env in ("1", "0", "1"):
    os.environ["ZKG16_SYNTH_THREADS"] = env
    t0 = time.perf_counter()
    h = C.c_void_p()
    rc = lib.zkg16_circuit_matrix(n, ones.reshape(-1), ones.reshape(-1), C.byref(h))
    t1 = time.perf_counter()
    c = circuits.SynthesizedCircuit(h)
    t2 = time.perf_counter()
    print("ZKG16_SYNTH_THREADS=%s  n=%d: build %.3f s, dims+alloc+export %.3f s, total %.3f s  (%d constraints, cpu count %d)" %
          (env, n, t1 - t0, t2 - t1, t2 - t0, c.num_constraints, os.cpu_count()), flush=True)
    del c
t0 = time.perf_counter()
z = circuits.matrix_witness(ones, ones, 8675313 + 4 if n == 128 else 0) if n == 128 else None
print("assignment only: %.3f s" % (time.perf_counter() - t0))
