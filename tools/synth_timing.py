"""Host-side synthesis breakdown (development probe): python tools/synth_timing.py [matrix_n]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import _lib
from zksnark_finalproject_amd.circuits import matrix_circuit
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a = np.ones(n * n, dtype=np.uint64)
for rep in range(3):
    h = C.c_void_p()
    t0 = time.perf_counter(); lib.zkg16_circuit_matrix(n, a, a, C.byref(h)); t1 = time.perf_counter()
    lib.zkg16_circuit_free(h); t2 = time.perf_counter()
    c = matrix_circuit(a.reshape(n, n), a.reshape(n, n)); t3 = time.perf_counter()
    print("n=%d synth %.4f free %.4f | python matrix_circuit (synth + dims + export + free) %.4f" % (n, t1 - t0, t2 - t1, t3 - t2), flush=True)
