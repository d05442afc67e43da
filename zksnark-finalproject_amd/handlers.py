"""In-process mirrors of the reference's request handlers around the hot path (the callers of `Groth16::prove`):
    prove_matrix     src/arkworks/backend/matrix_proof.rs:94-166   (POST /api/matrix_prove/prove)
    prove_fibonacci  src/arkworks/backend/fibbonaci_handler.rs:98-145
    prove_prime / verify_prime  src/arkworks/backend/prime_snark.rs:49-146, 165-206
Same steps, same response fields: synthesize the circuit (host C++ mirror), per-request Groth16 setup (on the device,
zkg16_setup), prove (zkg16_prove_resident), encode (wire.py).  The reference's HTTP layer (actix-web) is out of scope; the
trapdoor and r, s come from Python's PRNG rather than arkworks' StdRng stream, so proofs are valid Groth16 proofs for the
same statement but not the byte string the Rust server would emit for its seed."""
import random
import time

import numpy as np

from . import wire
from .circuits import fibonacci_circuit, fibonacci_circuit_handle, matrix_circuit, prime_circuit, prime_circuit_handle, prime_public_inputs, prime_search
from .workloads import R_MOD, g1_generator, g2_generator


def _fr_mont(x):
    v = (x << 256) % R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class _CachedShape:
    """What a request needs of a MatrixCircuit whose matrices are already on the device: the matrices depend on the size
    alone, so they are synthesized, exported and uploaded once per size and ctx; a request then only computes its assignment
    (zkg16_circuit_matrix_witness)."""

    def __init__(self, circ, rh):
        self.rh = rh
        self.num_instance, self.num_witness, self.num_vars = circ.num_instance, circ.num_witness, circ.num_vars
        self.num_constraints, self.domain = circ.num_constraints, circ.domain

    @classmethod
    def on_device(cls, dev, n):
        """The shape with its matrices written ON THE DEVICE (zkg16_r1cs_matrix): nothing of the circuit is synthesized on the host."""
        import ctypes as C
        from . import _lib
        nc, nw = C.c_size_t(), C.c_size_t()
        rc = _lib.load().zkg16_matrix_r1cs_dims(n, C.byref(nc), C.byref(nw), None)
        if rc:
            raise _lib.Zkg16Error(rc, "zkg16_matrix_r1cs_dims")
        self = cls.__new__(cls)
        self.rh = dev.r1cs_matrix(n)
        self.num_instance, self.num_witness, self.num_vars = 4, nw.value, 4 + nw.value
        self.num_constraints = nc.value
        self.domain = 1 << max(nc.value + 4 - 1, 0).bit_length()
        return self

    def instantiate(self, a, b):
        """The request's circuit without a host assignment: z is built on the device inside the proof (zkg16_prove_matrix)."""
        inst = _CachedShape.__new__(_CachedShape)
        inst.__dict__.update(self.__dict__)
        inst.z = None
        inst.matrices = (a, b)
        inst.public_inputs = None           # hash_a, hash_b, hash_c come back from the proof call
        inst.r1cs = None
        return inst


def _setup_and_prove(dev, circ, rng, keep_key=False):
    trap = np.stack([_fr_mont(rng.randrange(1, R_MOD)) for _ in range(5)])
    # arkworks draws random generators; any subgroup generator gives a valid key: [k]G for random k
    from .device import scalar_mul
    k = np.array([rng.getrandbits(62) for _ in range(4)], dtype=np.uint64)
    g1 = scalar_mul("g1", g1_generator(), k)[0]
    g2 = scalar_mul("g2", g2_generator(), k)[0]
    cached = getattr(circ, "rh", None) is not None
    direct = getattr(circ, "handle", None) is not None          # a CircuitHandle: matrices and assignment go to the device unexported
    wh_direct = None
    if direct:
        rh, wh_direct = dev.circuit_load(circ)
        circ.close()
    else:
        rh = circ.rh if cached else dev.r1cs_load(circ.r1cs, circ.num_vars)
    # cached shape: the assignment needs neither the key nor (for its host half, the three Poseidon chains) the device, so it is
    # started now and runs beside the setup's kernels (zkg16_witness_matrix takes the ctx only for its ~1 ms of kernels): at 128x128
    # the 61 ms of chains disappear under the 0.2 s setup and the proof is the plain resident one (0.17 s instead of 0.195 s streamed)
    early = None
    if circ.z is None and not keep_key and not direct:
        import threading
        early = {}

        def _assign():
            try:
                early["out"] = dev.witness_matrix(circ.matrices[0], circ.matrices[1])
            except Exception as e:      # noqa: BLE001 - re-raised on the caller's thread
                early["err"] = e
        early["thread"] = threading.Thread(target=_assign)
        early["thread"].start()
    t0 = time.perf_counter()
    try:
        if keep_key:        # tests want the key on the host as well
            pk, vk = dev.setup(rh, circ.num_instance, circ.num_vars, circ.domain, trap, g1, g2)
            ph = dev.pk_load(pk, circ.num_instance)
        else:               # the request path: the key never leaves the device
            pk = None
            ph, vk = dev.setup_resident(rh, circ.num_instance, trap, g1, g2)
    except Exception:
        if early is not None:           # the assignment thread must not outlive the request
            early["thread"].join()
            if "out" in early:
                dev.witness_free(early["out"][0])
        raise
    setup_time = time.perf_counter() - t0
    r, s = _fr_mont(rng.randrange(R_MOD)), _fr_mont(rng.randrange(R_MOD))
    if early is not None:               # cached shape, assignment started before the setup
        t0 = time.perf_counter()
        early["thread"].join()
        if "err" in early:
            dev.pk_free(ph)
            raise early["err"]
        wh, pub, _ = early["out"]
        proof, inf = dev.prove_resident(ph, rh, wh, r, s)
        proving_time = time.perf_counter() - t0
        circ.public_inputs = pub
    elif circ.z is None and not direct:  # cached shape: the assignment is produced on the device while the proof runs
        t0 = time.perf_counter()
        proof, inf, pub, _ = dev.prove_matrix(ph, rh, circ.matrices[0], circ.matrices[1], r, s)
        proving_time = time.perf_counter() - t0
        circ.public_inputs = pub
        wh = None
    else:
        wh = wh_direct if direct else dev.witness_load(circ.z)
        t0 = time.perf_counter()
        proof, inf = dev.prove_resident(ph, rh, wh, r, s)
        proving_time = time.perf_counter() - t0
    for f, h in ((dev.pk_free, ph),) + (((dev.witness_free, wh),) if wh is not None else ()) + (() if cached else ((dev.r1cs_free, rh),)):
        f(h)
    return dict(proof=proof, inf=inf, vk=vk, pk=pk, setup_time=setup_time, proving_time=proving_time, r=r, s=s)


def prove_matrix(dev, size, matrix_a, matrix_b, seed=0, keep_key=False):
    """-> the reference's ProveOutput fields (matrix_proof.rs:80-91)."""
    a = np.asarray(matrix_a, dtype=np.uint64).reshape(size, size)
    b = np.asarray(matrix_b, dtype=np.uint64).reshape(size, size)
    shapes = dev.__dict__.setdefault("_matrix_shapes", {})
    if keep_key or size < 2:            # tests want the host copy of everything
        circ = matrix_circuit(a, b)
    elif size in shapes:
        circ = shapes[size].instantiate(a, b)
    else:
        shapes[size] = _CachedShape.on_device(dev, size)           # first request of this size: matrices written by kernels
        circ = shapes[size].instantiate(a, b)
    out = _setup_and_prove(dev, circ, random.Random(seed), keep_key)
    ha, hb, hc = circ.public_inputs
    return dict(hash_a=wire.encode_hash(ha), hash_b=wire.encode_hash(hb), hash_c=wire.encode_hash(hc),
                setup_time=out["setup_time"], proving_time=out["proving_time"],
                # the reference counts matrix_mul twice (outer cs + circuit: matrix_proof.rs:108,150,160)
                num_constraints=circ.num_constraints + 2 * size ** 3, num_constraints_circuit=circ.num_constraints,
                num_variables=circ.num_instance, proof=wire.encode_proof(out["proof"], out["inf"]),
                # the reference returns the prepared key (encode_pvk, io.rs:62-68); the plain key is kept beside it
                pvk=wire.encode_pvk(out["vk"]), vk=wire.encode_vk(out["vk"]), _detail=out, _circuit=circ)


def prove_fibonacci(dev, a, b, num_of_rounds, seed=42, keep_key=False):
    """-> the reference's OutputDataFib-like fields (fibbonaci_handler.rs:84-90)."""
    circ = fibonacci_circuit(a, b, num_of_rounds) if keep_key else fibonacci_circuit_handle(a, b, num_of_rounds)
    out = _setup_and_prove(dev, circ, random.Random(seed), keep_key)
    return dict(proof=wire.encode_proof(out["proof"], out["inf"]), proving_time=out["proving_time"], setup_time=out["setup_time"],
                num_constraints=circ.num_constraints, num_variables=circ.num_instance,
                fib_number=[wire.encode_hash(x) for x in circ.public_inputs][-1], pvk=wire.encode_pvk(out["vk"]), vk=wire.encode_vk(out["vk"]),
                _detail=out, _circuit=circ)


def prove_prime(dev, x, i, seed=7, keep_key=False, check_satisfied=False):
    """-> the reference's ProveOutput fields of the prime handler (prime_snark.rs:36-47): search j in 0..=i for the first
    hash(x + j) mod 2^20 that passes the Fermat test, build PrimeCircuit for it, circuit-specific setup, prove."""
    found = prime_search(x, i)
    if not found["found"]:
        return dict(proof="", j=0, num_constraints=0, num_variables=0, setup_time=0.0, proving_time=0.0, found_prime=False, prime_num="", vk="")
    if keep_key or check_satisfied:     # tests want the arrays (and the satisfaction check) on the host
        circ = prime_circuit(x, found["j"], search=False, check_satisfied=check_satisfied)
    else:
        circ = prime_circuit_handle(x, found["j"])
    out = _setup_and_prove(dev, circ, random.Random(seed), keep_key)
    return dict(proof=wire.encode_proof(out["proof"], out["inf"]), j=found["j"], num_constraints=circ.num_constraints,
                num_variables=circ.num_vars, setup_time=out["setup_time"], proving_time=out["proving_time"], found_prime=True,
                prime_num=str(found["prime"]), pvk=wire.encode_pvk(out["vk"]), vk=wire.encode_vk(out["vk"]), satisfied=circ.satisfied,
                _detail=out, _circuit=circ)


def verify_prime(vk, x, j, proof_b64):
    """Mirror of verify_prime (prime_snark.rs:165-206): the reference re-synthesizes PrimeCircuit for (x, j) to recover the
    public inputs (x and the 256 digest bits); here they are computed natively (the same values: test_prime_circuit.py), then
    the proof is checked."""
    return verify_proof(vk, prime_public_inputs(x, j), proof_b64)


def verify_proof(vk, public_inputs_mont, proof_b64):
    """Mirror of the verify handlers (matrix_proof.rs:183-205, fibbonaci_handler.rs:118-145): decode the base64 compressed
    proof (and key, when given as the base64 string the prove mirrors return), check the Groth16 equation with the host
    verifier (zkg16_verify) -> {valid, verifying_time}."""
    from ._lib import Zkg16Error
    from .device import verify, verify_prepared
    t0 = time.perf_counter()
    try:
        if isinstance(vk, str):
            raw = __import__("base64").standard_b64decode(vk)
            n = int.from_bytes(raw[336:344], "little") if len(raw) >= 344 else 0
            vk = wire.pvk_deserialize_compressed(raw) if len(raw) > 344 + 48 * n else wire.vk_deserialize_compressed(raw)
        proof, inf = wire.decode_proof(proof_b64)
        t1 = time.perf_counter()
        ok = verify_prepared(vk, public_inputs_mont, proof, inf) if "alpha_beta" in vk else verify(vk, public_inputs_mont, proof, inf)
        t2 = time.perf_counter()
    except (ValueError, IndexError, Zkg16Error):
        # the reference's decode_proof / decode_pvk return None and the handler answers invalid; a key or input list of the
        # wrong shape is the same answer, never an exception out of the handler
        return dict(valid=False, verifying_time=0.0, decode_time=time.perf_counter() - t0)
    # verifying_time = the verification call alone, as the reference's timer (matrix_proof.rs:199-206, prime_snark.rs:191-200: started
    # after decode_pvk / decode_proof); decode_time = base64 + decompression + subgroup checks of key and proof (Python big ints)
    return dict(valid=bool(ok), verifying_time=t2 - t1, decode_time=t1 - t0)
