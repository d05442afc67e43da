"""In-process counterpart of the reference's sweep drivers (bench/matrix.py:10-60: all-ones matrices of size 2^k, prove
then verify; bench/fibo.py:26-60: Fibonacci rounds 0..186 with a = 0, b = 1), through the handler mirrors instead of
HTTP.  Records the reference's fields: num_constraints, setup_time, proving_time, verifying_time (seconds) of the SECOND request of
every size (the first one of a size in a process also builds that domain's NTT tables and grows the workspaces).
bench/prime.py:17-70: random x of 2, 4, .. 64 bits, i = 32 candidates, retried until a prime is found, prove then verify
with the returned pvk.
    python tools/sweep.py matrix [max_power=6] [warm]   |   python tools/sweep.py fib [step=31]   |   python tools/sweep.py prime [per_size=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zksnark_finalproject_amd import Device, handlers

dev = Device(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "matrix"
if kind == "matrix":
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    print("size,num_constraints,request_time,setup_time,proving_time,verifying_time,valid,decode_time")
    handlers.prove_matrix(dev, 2, np.ones((2, 2), dtype=np.uint64), np.ones((2, 2), dtype=np.uint64))      # warm-up (tables, streams)
    if len(sys.argv) > 3 and sys.argv[3] == "warm":      # a server that has served the largest size before: its buffers exist (on some
        nw = 1 << top                                    # boxes the first multi-GB hipMalloc calls of a process cost 1-3 s, once)
        handlers.prove_matrix(dev, nw, np.ones((nw, nw), dtype=np.uint64), np.ones((nw, nw), dtype=np.uint64))
    for k in range(1, top + 1):
        n = 1 << k
        ones = np.ones((n, n), dtype=np.uint64)
        if n < (1 << top) or not (len(sys.argv) > 3 and sys.argv[3] == "warm"):
            handlers.prove_matrix(dev, n, ones, ones, seed=100 + k)      # the first request of a size also builds that domain's NTT tables and
        t0 = time.perf_counter()                                       # grows the workspaces: the server's steady state is the second one
        res = handlers.prove_matrix(dev, n, ones, ones, seed=k)
        t1 = time.perf_counter()
        v = handlers.verify_proof(res["pvk"], res["_circuit"].public_inputs, res["proof"])      # the prepared key, as the reference's verify handler
        print("%d,%d,%.4f,%.4f,%.5f,%.5f,%s,%.5f" % (n, res["num_constraints_circuit"], t1 - t0, res["setup_time"], res["proving_time"],
                                               v["verifying_time"], v["valid"], v["decode_time"]), flush=True)
elif kind == "prime":
    import random
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rng = random.Random(2024)
    print("x_bits,x,j,prime_num,num_constraints,num_variables,request_time,setup_time,proving_time,verifying_time,valid,decode_time")
    for k in range(1, 7):
        bits = 1 << k
        for _ in range(per):
            while True:
                x = rng.randint(1 << (bits // 2), (1 << bits) - 1 - (32 if bits == 64 else 0))
                t0 = time.perf_counter()
                res = handlers.prove_prime(dev, x, 32)
                t1 = time.perf_counter()
                if res["found_prime"]:
                    break
            v = handlers.verify_prime(res["pvk"], x, res["j"], res["proof"])
            print("%d,%d,%d,%s,%d,%d,%.4f,%.4f,%.5f,%.5f,%s,%.5f" % (bits, x, res["j"], res["prime_num"], res["num_constraints"], res["num_variables"],
                                                             t1 - t0, res["setup_time"], res["proving_time"], v["verifying_time"], v["valid"], v["decode_time"]), flush=True)
else:
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    print("num_of_rounds,num_constraints,request_time,setup_time,proving_time,verifying_time,valid,decode_time")
    handlers.prove_fibonacci(dev, 0, 1, 5)
    for rounds in range(0, 187, step):
        t0 = time.perf_counter()
        res = handlers.prove_fibonacci(dev, 0, 1, rounds)
        t1 = time.perf_counter()
        v = handlers.verify_proof(res["pvk"], res["_circuit"].public_inputs, res["proof"])
        print("%d,%d,%.4f,%.4f,%.5f,%.5f,%s,%.5f" % (rounds, res["num_constraints"], t1 - t0, res["setup_time"], res["proving_time"],
                                               v["verifying_time"], v["valid"], v["decode_time"]), flush=True)
