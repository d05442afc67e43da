"""Host-side circuit synthesis through the C ABI (csrc/circuits.hip): the reference's MatrixCircuit (+ Poseidon) and
FibonacciCircuit as R1CS (CSR) + full assignment, ready for Device.r1cs_load / Device.witness_load.  No GPU needed."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Zkg16Error


class SynthesizedCircuit:
    """num_instance (incl. the constant 1), num_witness, num_constraints, r1cs (dict of CSR triples), z (Montgomery)."""

    def __init__(self, handle, check_satisfied=None):
        lib = _lib.load()
        ni, nw, nc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        nnz = (C.c_size_t * 3)()
        lib.zkg16_circuit_dims(handle, C.byref(ni), C.byref(nw), C.byref(nc), C.byref(nnz))
        self.num_instance, self.num_witness, self.num_constraints = ni.value, nw.value, nc.value
        self.num_vars = ni.value + nw.value
        self.satisfied = bool(lib.zkg16_circuit_is_satisfied(handle)) if ((nc.value <= 200000 and check_satisfied is not False) or check_satisfied) else None
        rp = [np.zeros(nc.value + 1, dtype=np.uint64) for _ in range(3)]
        col = [np.zeros(max(nnz[m], 1), dtype=np.uint32) for m in range(3)]
        cf = [np.zeros((max(nnz[m], 1), 4), dtype=np.uint64) for m in range(3)]
        z = np.zeros((self.num_vars, 4), dtype=np.uint64)
        arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
        rc = lib.zkg16_circuit_export(handle, C.byref(arr(rp)), C.byref(arr(col)), C.byref(arr(cf)), z)
        if rc:
            raise Zkg16Error(rc, "circuit export")
        self.r1cs = dict(a=(rp[0], col[0][:nnz[0]], cf[0][:nnz[0]]), b=(rp[1], col[1][:nnz[1]], cf[1][:nnz[1]]),
                         c=(rp[2], col[2][:nnz[2]], cf[2][:nnz[2]]), num_inputs=self.num_instance, num_constraints=self.num_constraints)
        self.z = z
        self.public_inputs = z[1:self.num_instance].copy()
        self.domain = 1 << max(self.num_constraints + self.num_instance - 1, 0).bit_length()
        lib.zkg16_circuit_free(handle)


class CircuitHandle:
    """A synthesized circuit that stays inside the library: its dimensions and public inputs only.  Device.circuit_load(self) puts
    its matrices and assignment on the device without exporting them (the request path of circuits that are re-synthesized per
    request); close() (or garbage collection) frees it."""

    def __init__(self, handle):
        lib = _lib.load()
        self.handle = handle
        ni, nw, nc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        nnz = (C.c_size_t * 3)()
        lib.zkg16_circuit_dims(handle, C.byref(ni), C.byref(nw), C.byref(nc), C.byref(nnz))
        self.num_instance, self.num_witness, self.num_constraints = ni.value, nw.value, nc.value
        self.num_vars = ni.value + nw.value
        pub = np.zeros((max(ni.value - 1, 0), 4), dtype=np.uint64)
        if ni.value > 1:
            rc = lib.zkg16_circuit_public_inputs(handle, pub.reshape(-1), ni.value - 1)
            if rc:
                raise Zkg16Error(rc, "zkg16_circuit_public_inputs")
        self.public_inputs = pub
        self.domain = 1 << max(self.num_constraints + self.num_instance - 1, 0).bit_length()
        self.satisfied = None
        self.z = self.r1cs = None

    def close(self):
        if self.handle is not None:
            _lib.load().zkg16_circuit_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass


def prime_public_inputs(x, j):
    """The PrimeCircuit's public inputs for candidate j (x and the 256 digest bits), computed natively (zkg16_prime_public_inputs)."""
    out = np.zeros((257, 4), dtype=np.uint64)
    rc = _lib.load().zkg16_prime_public_inputs(x, j, out.reshape(-1))
    if rc:
        raise Zkg16Error(rc, "zkg16_prime_public_inputs")
    return out


def prime_circuit_handle(x, j):
    """PrimeCircuit of candidate j as a CircuitHandle (nothing exported to Python)."""
    h = C.c_void_p()
    rc = _lib.load().zkg16_circuit_prime(x, j, C.byref(h))
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_prime")
    c = CircuitHandle(h)
    c.j = j
    return c


def matrix_circuit(a, b):
    """MatrixCircuit for u64 matrices a, b (n x n lists/arrays); public inputs = Poseidon hashes of A, B, C = A*B."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    n = a.shape[0]
    assert a.shape == (n, n) and b.shape == (n, n)
    h = C.c_void_p()
    rc = _lib.load().zkg16_circuit_matrix(n, a.reshape(-1), b.reshape(-1), C.byref(h))
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_matrix")
    return SynthesizedCircuit(h)


def matrix_witness(a, b, num_vars):
    """Only the full assignment of matrix_circuit(a, b) (zkg16_circuit_matrix_witness): for callers that kept the matrices of
    this size.  num_vars = that circuit's num_instance + num_witness."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    n = a.shape[0]
    z = np.zeros((num_vars, 4), dtype=np.uint64)
    rc = _lib.load().zkg16_circuit_matrix_witness(n, a.reshape(-1), b.reshape(-1), z.reshape(-1), num_vars)
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_matrix_witness")
    return z


def matrix_sponge_states(a, b):
    """The host half of the device witness generator (zkg16_matrix_sponge_states; no GPU): the three native sponges of the
    matrix handler with the state in front of every permutation -> (states [3, ceil(n^2/2), 3, 4], hashes [3, 4])."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    n = a.shape[0]
    perms = (n * n + 1) // 2
    states = np.zeros((3, perms, 3, 4), dtype=np.uint64)
    hashes = np.zeros((3, 4), dtype=np.uint64)
    rc = _lib.load().zkg16_matrix_sponge_states(n, a.reshape(-1), b.reshape(-1), states.ctypes.data, hashes.reshape(-1))
    if rc:
        raise Zkg16Error(rc, "zkg16_matrix_sponge_states")
    return states, hashes


def matrix_r1cs_from_plan(n):
    """The MatrixCircuit's R1CS of size n from its plan (zkg16_matrix_r1cs_dims / _host: templates + closed forms, host loops) ->
    (r1cs dict as SynthesizedCircuit.r1cs, num_witness).  What the device kernel of zkg16_r1cs_matrix is tested against."""
    lib = _lib.load()
    nc, nw = C.c_size_t(), C.c_size_t()
    nnz = (C.c_size_t * 3)()
    rc = lib.zkg16_matrix_r1cs_dims(n, C.byref(nc), C.byref(nw), C.byref(nnz))
    if rc:
        raise Zkg16Error(rc, "zkg16_matrix_r1cs_dims")
    rp = [np.zeros(nc.value + 1, dtype=np.uint64) for _ in range(3)]
    col = [np.zeros(max(nnz[m], 1), dtype=np.uint32) for m in range(3)]
    cf = [np.zeros((max(nnz[m], 1), 4), dtype=np.uint64) for m in range(3)]
    arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
    rc = lib.zkg16_matrix_r1cs_host(n, C.byref(arr(rp)), C.byref(arr(col)), C.byref(arr(cf)))
    if rc:
        raise Zkg16Error(rc, "zkg16_matrix_r1cs_host")
    return dict(a=(rp[0], col[0][:nnz[0]], cf[0][:nnz[0]]), b=(rp[1], col[1][:nnz[1]], cf[1][:nnz[1]]),
                c=(rp[2], col[2][:nnz[2]], cf[2][:nnz[2]]), num_inputs=4, num_constraints=nc.value), nw.value


def fibonacci_circuit(a, b, steps):
    h = C.c_void_p()
    rc = _lib.load().zkg16_circuit_fibonacci(a, b, steps, C.byref(h))
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_fibonacci")
    return SynthesizedCircuit(h)


def fibonacci_circuit_handle(a, b, steps):
    """FibonacciCircuit as a CircuitHandle (nothing exported to Python)."""
    h = C.c_void_p()
    rc = _lib.load().zkg16_circuit_fibonacci(a, b, steps, C.byref(h))
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_fibonacci")
    return CircuitHandle(h)


def prime_search(x, i_max):
    """The prime handler's native search (backend/prime_snark.rs:57-70): first j <= i_max whose hash(x + j) mod 2^20 passes the
    Fermat test -> dict(found, j, prime, digest) (zkg16_prime_search)."""
    j, p, found = C.c_uint64(0), C.c_uint32(0), C.c_int(0)
    digest = np.zeros(32, dtype=np.uint8)
    rc = _lib.load().zkg16_prime_search(x, i_max, C.byref(j), C.byref(p), digest, C.byref(found))
    if rc:
        raise Zkg16Error(rc, "zkg16_prime_search")
    return dict(found=bool(found.value), j=j.value, prime=p.value, digest=bytes(digest))


def prime_candidate(x, j):
    """check_if_next_is_prime(x, j) natively -> dict(digest, n, bases, r_bytes, is_prime)."""
    n, isp = C.c_uint32(0), C.c_int(0)
    digest, rb = np.zeros(32, dtype=np.uint8), np.zeros(32, dtype=np.uint8)
    bases = np.zeros(3, dtype=np.uint32)
    rc = _lib.load().zkg16_prime_candidate(x, j, digest, C.byref(n), bases, rb, C.byref(isp))
    if rc:
        raise Zkg16Error(rc, "zkg16_prime_candidate")
    return dict(digest=bytes(digest), n=n.value, bases=[int(b) for b in bases], r_bytes=bytes(rb), is_prime=bool(isp.value))


def prime_circuit(x, i_max_or_j, search=True, check_satisfied=True):
    """The reference's PrimeCircuit (C++ mirror).  search=True: as prove_prime — find the first prime candidate j <= i_max and
    build its circuit (raises if none); search=False: the circuit of candidate j itself, as verify_prime rebuilds it.
    check_satisfied: evaluate all 338 k constraints on the assignment (a test convenience — neither `Groth16::prove` nor the
    reference's handlers do it; it costs about as much as the synthesis itself, so timed paths pass False)."""
    j = i_max_or_j
    if search:
        res = prime_search(x, i_max_or_j)
        if not res["found"]:
            raise ValueError("no prime candidate for x=%d within i=%d" % (x, i_max_or_j))
        j = res["j"]
    h = C.c_void_p()
    rc = _lib.load().zkg16_circuit_prime(x, j, C.byref(h))
    if rc:
        raise Zkg16Error(rc, "zkg16_circuit_prime")
    c = SynthesizedCircuit(h, check_satisfied=check_satisfied)
    c.j = j
    return c


def poseidon_hash(elems_mont):
    e = np.ascontiguousarray(elems_mont, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    rc = _lib.load().zkg16_poseidon_hash(e, e.shape[0], out)
    if rc:
        raise Zkg16Error(rc, "zkg16_poseidon_hash")
    return out
