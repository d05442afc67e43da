"""The PrimeCircuit host synthesis in a tight loop, sequential against the per-part threads (development probe)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zksnark_finalproject_amd import _lib
from zksnark_finalproject_amd.circuits import prime_search
lib = _lib.load()
x = 58405
f = prime_search(x, 32)
for mode in ("0", "1", "0", "1"):
    os.environ["ZKG16_SYNTH_THREADS"] = mode
    ts = []
    for it in range(20):
        t0 = time.perf_counter()
        h = C.c_void_p()
        lib.zkg16_circuit_prime(x, f["j"], C.byref(h))
        t1 = time.perf_counter()
        lib.zkg16_circuit_free(h)
        ts.append((t1 - t0) * 1e3)
    print("ZKG16_SYNTH_THREADS=%s: min %.1f median %.1f max %.1f ms" % (mode, min(ts), sorted(ts)[10], max(ts)), flush=True)
