"""Where a prime request's time goes (development probe)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers, wire
from zksnark_finalproject_amd.circuits import prime_circuit, prime_search
dev = Device(0)
handlers.prove_prime(dev, 58405, 32)
for x in (58405, 93, 13, 1234567):
    T = {}
    t = time.perf_counter()
    def lap(k):
        global t
        n = time.perf_counter(); T[k] = (n - t) * 1e3; t = n
    f = prime_search(x, 32); lap("search")
    circ = prime_circuit(x, f["j"], search=False, check_satisfied=False); lap("synthesis")
    rh = dev.r1cs_load(circ.r1cs, circ.num_vars); lap("r1cs_load")
    wh = dev.witness_load(circ.z); lap("witness_load")
    rng = random.Random(7)
    trap = np.stack([handlers._fr_mont(rng.randrange(1, handlers.R_MOD)) for _ in range(5)])
    from zksnark_finalproject_amd.device import scalar_mul
    from zksnark_finalproject_amd.workloads import g1_generator, g2_generator
    k = np.array([rng.getrandbits(62) for _ in range(4)], dtype=np.uint64)
    g1 = scalar_mul("g1", g1_generator(), k)[0]; g2 = scalar_mul("g2", g2_generator(), k)[0]; lap("generators")
    ph, vk = dev.setup_resident(rh, circ.num_instance, trap, g1, g2); lap("setup")
    r, s = handlers._fr_mont(5), handlers._fr_mont(9)
    proof, inf = dev.prove_resident(ph, rh, wh, r, s); lap("prove")
    dev.pk_free(ph); dev.witness_free(wh); dev.r1cs_free(rh); lap("free")
    e1 = wire.encode_proof(proof, inf); lap("encode_proof")
    e2 = wire.encode_pvk(vk); lap("encode_pvk")
    e3 = wire.encode_vk(vk); lap("encode_vk")
    print(x, "total %.1f ms: " % sum(T.values()) + ", ".join("%s %.1f" % kv for kv in T.items()), flush=True)
for x in (58405, 93, 13, 1234567):
    t0 = time.perf_counter()
    res = handlers.prove_prime(dev, x, 32)
    t1 = time.perf_counter()
    v = handlers.verify_prime(res["pvk"], x, res["j"], res["proof"])
    t2 = time.perf_counter()
    print("handler: prove_prime %.1f ms (setup %.1f, proof %.1f), verify_prime %.1f ms (verifying %.2f, decode %.1f) valid %s" % ((t1 - t0) * 1e3, res["setup_time"] * 1e3,
          res["proving_time"] * 1e3, (t2 - t1) * 1e3, v["verifying_time"] * 1e3, v["decode_time"] * 1e3, v["valid"]), flush=True)
