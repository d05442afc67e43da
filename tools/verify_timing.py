"""Host verifier timings (development probe): python tools/verify_timing.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from zksnark_finalproject_amd import Device, handlers
from zksnark_finalproject_amd.device import verify, pairing_check, point_check, pvk_prepare, verify_prepared
dev = Device(0)
res = handlers.prove_fibonacci(dev, 0, 1, 30)
vk = res["_detail"]["vk"]; circ = res["_circuit"]; proof, inf = res["_detail"]["proof"], res["_detail"]["inf"]
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("verify            %.2f ms" % t(lambda: verify(vk, circ.public_inputs, proof, inf)))
pvk = pvk_prepare(vk)
print("verify_prepared   %.2f ms" % t(lambda: verify_prepared(pvk, circ.public_inputs, proof, inf)))
g1 = np.stack([proof[0:12], proof[36:48]]); g2 = np.stack([proof[12:36], proof[12:36]])
print("pairing_check 1   %.2f ms" % t(lambda: pairing_check(g1[:1], g2[:1])))
print("pairing_check 2   %.2f ms" % t(lambda: pairing_check(g1, g2)))
print("point_check g1    %.2f ms" % t(lambda: point_check("g1", proof[0:12])))
print("point_check g2    %.2f ms" % t(lambda: point_check("g2", proof[12:36])))
