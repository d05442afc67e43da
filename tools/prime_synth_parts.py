"""The PrimeCircuit host synthesis under different surroundings (development probe)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import _lib
from zksnark_finalproject_amd.circuits import prime_search
lib = _lib.load()
def build(x, j):
    t0 = time.perf_counter()
    h = C.c_void_p()
    lib.zkg16_circuit_prime(x, j, C.byref(h))
    return h, (time.perf_counter() - t0) * 1e3
def export(h):
    ni, nw, nc = C.c_size_t(), C.c_size_t(), C.c_size_t(); nnz = (C.c_size_t * 3)()
    lib.zkg16_circuit_dims(h, C.byref(ni), C.byref(nw), C.byref(nc), C.byref(nnz))
    rp = [np.zeros(nc.value + 1, dtype=np.uint64) for _ in range(3)]
    col = [np.zeros(max(nnz[m], 1), dtype=np.uint32) for m in range(3)]
    cf = [np.zeros((max(nnz[m], 1), 4), dtype=np.uint64) for m in range(3)]
    z = np.zeros((ni.value + nw.value, 4), dtype=np.uint64)
    arr = lambda xs: (C.c_void_p * 3)(*[a.ctypes.data for a in xs])
    lib.zkg16_circuit_export(h, C.byref(arr(rp)), C.byref(arr(col)), C.byref(arr(cf)), z)
    return rp, col, cf, z
xs = [58405 + 977 * i for i in range(40)]
js = [prime_search(x, 32) for x in xs]
pairs = [(x, f["j"]) for x, f in zip(xs, js) if f["found"]][:16]
def show(tag, ts):
    print("%-40s min %.1f median %.1f max %.1f" % (tag, min(ts), sorted(ts)[len(ts) // 2], max(ts)), flush=True)
for rep in range(2):
    ts = []
    for x, j in pairs:
        h, t = build(pairs[0][0], pairs[0][1]); lib.zkg16_circuit_free(h); ts.append(t)
    show("A same x, tight", ts)
    ts = []
    for x, j in pairs:
        h, t = build(x, j); lib.zkg16_circuit_free(h); ts.append(t)
    show("B different x, tight", ts)
    ts = []
    for x, j in pairs:
        h, t = build(pairs[0][0], pairs[0][1]); keep = export(h); lib.zkg16_circuit_free(h); ts.append(t)
    show("C same x + numpy export", ts)
    ts = []
    for x, j in pairs:
        time.sleep(0.05)
        h, t = build(pairs[0][0], pairs[0][1]); lib.zkg16_circuit_free(h); ts.append(t)
    show("D same x, 50 ms sleep between", ts)
    ts = []
    junk = np.zeros(1 << 25, dtype=np.uint64)
    for x, j in pairs:
        junk += 1                    # 256 MB of traffic through the caches
        h, t = build(pairs[0][0], pairs[0][1]); lib.zkg16_circuit_free(h); ts.append(t)
    show("E same x, caches flushed between", ts)
