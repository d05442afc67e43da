"""Prime requests back to back through the handler mirror (development probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zksnark_finalproject_amd import Device, handlers
dev = Device(0)
for x in (58405, 93, 13, 1234567, 777, 31337, 2**40 + 5, 12):
    t0 = time.perf_counter()
    res = handlers.prove_prime(dev, x, 32)
    t1 = time.perf_counter()
    v = handlers.verify_prime(res["pvk"], x, res["j"], res["proof"])
    t2 = time.perf_counter()
    print("prove_prime %.1f ms (setup %.1f, proof %.1f), verify_prime %.1f ms valid %s" % ((t1 - t0) * 1e3, res["setup_time"] * 1e3, res["proving_time"] * 1e3, (t2 - t1) * 1e3, v["valid"]), flush=True)
