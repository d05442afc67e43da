"""Thin object layer over the C ABI: one `Device` == one zkg16_ctx == one MI355X."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Zkg16Error


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _opt_u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


class Device:
    def __init__(self, device_id=0):
        self.lib = _lib.load()
        self.ctx = C.c_void_p()
        ids = (C.c_int * 1)(device_id)
        rc = self.lib.zkg16_init(ids, 1, C.byref(self.ctx))
        if rc != 0:
            raise Zkg16Error(rc, self.lib.zkg16_strerror(rc).decode())

    def close(self):
        if self.ctx:
            self.lib.zkg16_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            detail = self.lib.zkg16_last_error(self.ctx).decode()
            raise Zkg16Error(rc, self.lib.zkg16_strerror(rc).decode() + (" — " + detail if detail else ""))

    # ---- residency
    def pk_load(self, pk, num_instance, shard_index=0, shard_count=1):
        """pk: dict of numpy arrays (see groth16.ProvingKey.as_arrays)."""
        a, b1, b2 = _u64(pk["a_query"]).reshape(-1, 12), _u64(pk["b_g1_query"]).reshape(-1, 12), _u64(pk["b_g2_query"]).reshape(-1, 24)
        h, l = _u64(pk["h_query"]).reshape(-1, 12), _u64(pk["l_query"]).reshape(-1, 12)
        infs = [_opt_u8(pk.get(k)) for k in ("a_inf", "b_g1_inf", "b_g2_inf", "h_inf", "l_inf")]
        handle = C.c_uint64()
        self._check(self.lib.zkg16_pk_load(
            self.ctx, a, _ptr(infs[0]), a.shape[0], b1, _ptr(infs[1]), b1.shape[0], b2, _ptr(infs[2]), b2.shape[0],
            _ptr(h), _ptr(infs[3]), h.shape[0], _ptr(l), _ptr(infs[4]), l.shape[0],
            _u64(pk["alpha_g1"]), _u64(pk["beta_g1"]), _u64(pk["beta_g2"]), _u64(pk["delta_g1"]), _u64(pk["delta_g2"]),
            num_instance, shard_index, shard_count, C.byref(handle)))
        return handle.value

    def pk_load_range(self, pk, num_instance, z_lo, z_hi, h_lo, h_hi, blinding):
        """A shard given by explicit index ranges (zkg16_pk_load_range); see shard_plan."""
        a, b1, b2 = _u64(pk["a_query"]).reshape(-1, 12), _u64(pk["b_g1_query"]).reshape(-1, 12), _u64(pk["b_g2_query"]).reshape(-1, 24)
        h, l = _u64(pk["h_query"]).reshape(-1, 12), _u64(pk["l_query"]).reshape(-1, 12)
        infs = [_opt_u8(pk.get(k)) for k in ("a_inf", "b_g1_inf", "b_g2_inf", "h_inf", "l_inf")]
        handle = C.c_uint64()
        self._check(self.lib.zkg16_pk_load_range(
            self.ctx, a, _ptr(infs[0]), a.shape[0], b1, _ptr(infs[1]), b1.shape[0], b2, _ptr(infs[2]), b2.shape[0],
            _ptr(h), _ptr(infs[3]), h.shape[0], _ptr(l), _ptr(infs[4]), l.shape[0],
            _u64(pk["alpha_g1"]), _u64(pk["beta_g1"]), _u64(pk["beta_g2"]), _u64(pk["delta_g1"]), _u64(pk["delta_g2"]),
            num_instance, z_lo, z_hi, h_lo, h_hi, int(bool(blinding)), C.byref(handle)))
        return handle.value

    def pk_precompute(self, pk_h, window_bits_z=0, window_bits_h=0):
        """Window tables for a key that stays resident (zkg16_pk_precompute): same proofs, fewer bucket additions.
        0 = width chosen from the query length, < 0 = leave that side without a table.  -> HBM bytes added."""
        added = C.c_uint64(0)
        self._check(self.lib.zkg16_pk_precompute(self.ctx, pk_h, int(window_bits_z), int(window_bits_h), C.byref(added)))
        return added.value

    def last_term_counts(self):
        """-> (z list, B list (0 = the z list was used), h list): mixed additions per MSM of the last proof (zkg16_last_term_counts)."""
        out = (C.c_uint64 * 3)()
        self._check(self.lib.zkg16_last_term_counts(self.ctx, out))
        return int(out[0]), int(out[1]), int(out[2])

    def lane_log(self, rows=64):
        """-> list of (lane, start_ms, end_ms) of the most recent proofs on this ctx (zkg16_lane_log), oldest first."""
        buf = (C.c_double * (3 * rows))()
        n = self.lib.zkg16_lane_log(self.ctx, buf, rows)
        return [(int(buf[3 * i]), float(buf[3 * i + 1]), float(buf[3 * i + 2])) for i in range(n)]

    def last_acc_waves(self):
        """-> G1 accumulation waves per SIMD of the last proof's (z, B, h) term lists (zkg16_last_acc_waves); 0 = list not built."""
        out = (C.c_int * 3)()
        self._check(self.lib.zkg16_last_acc_waves(self.ctx, out))
        return int(out[0]), int(out[1]), int(out[2])

    def circuit_load(self, circuit):
        """circuit: a circuits.CircuitHandle (a synthesized circuit still held by the library) -> (r1cs_handle, witness_handle), loaded
        without exporting its arrays to Python (zkg16_circuit_load)."""
        rh, wh = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.zkg16_circuit_load(self.ctx, circuit.handle, C.byref(rh), C.byref(wh)))
        return rh.value, wh.value

    def acc_resident_waves(self):
        """-> (G1, G2) waves of the accumulation kernels one SIMD holds at once (zkg16_acc_resident_waves)."""
        out = (C.c_int * 2)()
        self._check(self.lib.zkg16_acc_resident_waves(self.ctx, out))
        return int(out[0]), int(out[1])

    def pk_table_bits(self, pk_h):
        """-> (window bits of the z-side tables, of the h-side table); 0 = none (zkg16_pk_table_bits)."""
        bz, bh = C.c_int(0), C.c_int(0)
        self._check(self.lib.zkg16_pk_table_bits(self.ctx, pk_h, C.byref(bz), C.byref(bh)))
        return bz.value, bh.value

    def pk_slice(self, pk_h, z_lo, z_hi, h_lo, h_hi, blinding):
        """A shard cut out of a whole resident key, device to device (zkg16_pk_slice)."""
        handle = C.c_uint64()
        self._check(self.lib.zkg16_pk_slice(self.ctx, pk_h, z_lo, z_hi, h_lo, h_hi, int(bool(blinding)), C.byref(handle)))
        return handle.value

    def pk_free(self, h):
        self.lib.zkg16_pk_free(self.ctx, h)

    @staticmethod
    def _csr(r1cs):
        args, keep = [], []
        for m in ("a", "b", "c"):
            rp, col, cf = r1cs[m]
            rp = _u64(rp)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            cf = _u64(cf).reshape(-1, 4)
            keep += [rp, col, cf]
            args += [rp, _ptr(col) if col.size else None, _ptr(cf) if cf.size else None]
        return args, keep

    def r1cs_load(self, r1cs, num_variables):
        args, keep = self._csr(r1cs)
        handle = C.c_uint64()
        self._check(self.lib.zkg16_r1cs_load(self.ctx, *args, r1cs["num_inputs"], r1cs["num_constraints"], num_variables, C.byref(handle)))
        return handle.value

    def r1cs_matrix(self, n):
        """The MatrixCircuit's R1CS of size n written on the device (zkg16_r1cs_matrix) -> r1cs handle."""
        handle = C.c_uint64()
        self._check(self.lib.zkg16_r1cs_matrix(self.ctx, n, C.byref(handle)))
        return handle.value

    def r1cs_read(self, h):
        """-> the r1cs dict (as SynthesizedCircuit.r1cs) + num_variables behind a handle (zkg16_r1cs_read)."""
        ni, nc, nv = C.c_size_t(), C.c_size_t(), C.c_size_t()
        nnz = (C.c_size_t * 3)()
        self._check(self.lib.zkg16_r1cs_read(self.ctx, h, None, None, None, C.byref(ni), C.byref(nc), C.byref(nv), C.byref(nnz)))
        rp = [np.zeros(nc.value + 1, dtype=np.uint64) for _ in range(3)]
        col = [np.zeros(max(nnz[m], 1), dtype=np.uint32) for m in range(3)]
        cf = [np.zeros((max(nnz[m], 1), 4), dtype=np.uint64) for m in range(3)]
        arr = lambda xs: (C.c_void_p * 3)(*[x.ctypes.data for x in xs])
        a, b, c = arr(rp), arr(col), arr(cf)
        self._check(self.lib.zkg16_r1cs_read(self.ctx, h, C.addressof(a), C.addressof(b), C.addressof(c), None, None, None, None))
        return dict(a=(rp[0], col[0][:nnz[0]], cf[0][:nnz[0]]), b=(rp[1], col[1][:nnz[1]], cf[1][:nnz[1]]),
                    c=(rp[2], col[2][:nnz[2]], cf[2][:nnz[2]]), num_inputs=ni.value, num_constraints=nc.value), nv.value

    def r1cs_free(self, h):
        self.lib.zkg16_r1cs_free(self.ctx, h)

    def witness_load(self, z):
        z = _u64(z).reshape(-1, 4)
        handle = C.c_uint64()
        self._check(self.lib.zkg16_witness_load(self.ctx, z, z.shape[0], C.byref(handle)))
        return handle.value

    def witness_matrix(self, a, b):
        """The MatrixCircuit's assignment for (a, b) built on the device (zkg16_witness_matrix) ->
        (witness handle, public inputs [3, 4] = hash_a, hash_b, hash_c, dict of ms: host sponges / device / whole call)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        n = a.shape[0]
        if a.shape != (n, n) or b.shape != (n, n):
            raise ValueError("witness_matrix: a and b must be n x n")
        handle = C.c_uint64()
        pub = np.zeros((3, 4), dtype=np.uint64)
        ms = (C.c_float * 3)()
        self._check(self.lib.zkg16_witness_matrix(self.ctx, n, a.reshape(-1), b.reshape(-1), C.byref(handle), pub.ctypes.data, C.addressof(ms)))
        return handle.value, pub, dict(host_sponges_ms=float(ms[0]), device_ms=float(ms[1]), call_ms=float(ms[2]))

    def witness_read(self, h, n_assign):
        z = np.zeros((n_assign, 4), dtype=np.uint64)
        self._check(self.lib.zkg16_witness_read(self.ctx, h, z.reshape(-1), n_assign))
        return z

    def witness_free(self, h):
        self.lib.zkg16_witness_free(self.ctx, h)

    # ---- proofs
    def prove_resident(self, pk_h, r1cs_h, wit_h, r, s):
        proof = np.zeros(48, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        self._check(self.lib.zkg16_prove_resident(self.ctx, pk_h, r1cs_h, wit_h, _u64(r), _u64(s), proof, inf))
        return proof, inf

    def prove_matrix(self, pk_h, r1cs_h, a, b, r, s):
        """One matrix-handler request on resident matrices: assignment built on the device while the proof already runs
        (zkg16_prove_matrix) -> (proof, inf, public inputs [3, 4], dict of ms)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        n = a.shape[0]
        if a.shape != (n, n) or b.shape != (n, n):
            raise ValueError("prove_matrix: a and b must be n x n")
        proof = np.zeros(48, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        pub = np.zeros((3, 4), dtype=np.uint64)
        ms = (C.c_float * 3)()
        self._check(self.lib.zkg16_prove_matrix(self.ctx, pk_h, r1cs_h, n, a.reshape(-1), b.reshape(-1), _u64(r), _u64(s), proof, inf,
                                                pub.ctypes.data, C.addressof(ms)))
        return proof, inf, pub, dict(host_sponges_ms=float(ms[0]), parts=int(ms[1]), call_ms=float(ms[2]))

    def prove(self, pk_h, r, s, r1cs, z):
        args, keep = self._csr(r1cs)
        z = _u64(z).reshape(-1, 4)
        proof = np.zeros(48, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        self._check(self.lib.zkg16_prove(self.ctx, pk_h, _u64(r), _u64(s), *args, r1cs["num_inputs"], r1cs["num_constraints"],
                                         z, z.shape[0], proof, inf))
        return proof, inf

    def prove_partial(self, pk_h, r1cs_h, wit_h, r, s):
        part = np.zeros(72, dtype=np.uint64)
        inf = np.zeros(5, dtype=np.uint8)
        self._check(self.lib.zkg16_prove_partial(self.ctx, pk_h, r1cs_h, wit_h, _u64(r), _u64(s), part, inf))
        return part, inf

    def prove_finish(self, pk_h, r, s, partials, partial_inf):
        partials = _u64(partials).reshape(-1, 72)
        partial_inf = np.ascontiguousarray(partial_inf, dtype=np.uint8).reshape(-1, 5)
        proof = np.zeros(48, dtype=np.uint64)
        inf = np.zeros(3, dtype=np.uint8)
        self._check(self.lib.zkg16_prove_finish(self.ctx, pk_h, _u64(r), _u64(s), partials, partial_inf, partials.shape[0], proof, inf))
        return proof, inf

    # ---- stages
    def ntt(self, data, inverse=False, coset=False):
        d = _u64(data).reshape(-1, 4).copy()
        n = d.shape[0]
        log_n = n.bit_length() - 1
        if 1 << log_n != n:
            raise ValueError("ntt: length %d is not a power of two" % n)
        self._check(self.lib.zkg16_ntt(self.ctx, d, log_n, int(inverse), int(coset)))
        return d

    def msm(self, group, bases, scalars_canonical, inf=None):
        w = 12 if group == "g1" else 24
        bases = _u64(bases).reshape(-1, w)
        sc = _u64(scalars_canonical).reshape(-1, 4)
        n = min(bases.shape[0], sc.shape[0])
        inf = _opt_u8(inf)
        out = np.zeros(w, dtype=np.uint64)
        oinf = np.zeros(1, dtype=np.uint8)
        fn = self.lib.zkg16_msm_g1 if group == "g1" else self.lib.zkg16_msm_g2
        self._check(fn(self.ctx, _ptr(bases) if n else None, _ptr(inf), _ptr(sc) if n else None, n, out, oinf))
        return out, int(oinf[0])

    def bench_msm(self, group, bases, scalars_canonical, iters=3, inf=None):
        w = 12 if group == "g1" else 24
        bases = _u64(bases).reshape(-1, w)
        sc = _u64(scalars_canonical).reshape(-1, 4)
        n = min(bases.shape[0], sc.shape[0])
        inf = _opt_u8(inf)
        out = np.zeros(w, dtype=np.uint64)
        oinf = np.zeros(1, dtype=np.uint8)
        ms = C.c_float()
        self._check(self.lib.zkg16_bench_msm(self.ctx, 1 if group == "g1" else 2, _ptr(bases), _ptr(inf), _ptr(sc), n, iters,
                                             C.byref(ms), out, oinf))
        return ms.value, out, int(oinf[0])

    def bench_ntt(self, log_n, inverse=False, coset=False, iters=10):
        ms = C.c_float()
        self._check(self.lib.zkg16_bench_ntt(self.ctx, log_n, int(inverse), int(coset), iters, C.byref(ms)))
        return ms.value

    def bench_witness_map(self, r1cs_h, wit_h, iters=3):
        """ms per stand-alone witness map (3 SpMV + 7 NTT + point-wise) on resident inputs."""
        ms = C.c_float()
        self._check(self.lib.zkg16_bench_witness_map(self.ctx, r1cs_h, wit_h, iters, C.byref(ms)))
        return ms.value

    def witness_map(self, r1cs_h, wit_h, n_max):
        h = np.zeros((n_max, 4), dtype=np.uint64)
        log_n = C.c_size_t()
        self._check(self.lib.zkg16_witness_map(self.ctx, r1cs_h, wit_h, h, C.byref(log_n)))
        return h[: 1 << log_n.value]

    def fixed_base(self, group, base, scalars_canonical):
        w = 12 if group == "g1" else 24
        sc = _u64(scalars_canonical).reshape(-1, 4)
        n = sc.shape[0]
        out = np.zeros((n, w), dtype=np.uint64)
        oinf = np.zeros(n, dtype=np.uint8)
        fn = self.lib.zkg16_fixed_base_g1 if group == "g1" else self.lib.zkg16_fixed_base_g2
        self._check(fn(self.ctx, _u64(base), _ptr(sc), n, _ptr(out), _ptr(oinf)))
        return out, oinf

    # ---- instrumentation
    def last_timings(self):
        buf = (C.c_float * 22)()
        n = self.lib.zkg16_last_timings(self.ctx, buf, 22)
        names = ["spmv", "witness_map", "msm_sort", "msm_h", "msm_l", "msm_a", "msm_b1", "msm_b2", "host_tail", "total_wall"]
        out = {names[i]: float(buf[i]) for i in range(min(n, 10))}
        if n >= 20:
            # device times under the span names of upstream's prover (ark-groth16 prover.rs): accumulate + fix-ups / bucket reduction
            acc = dict(zip("HLA", [float(buf[10]), float(buf[11]), float(buf[12])]), B1=float(buf[13]), B2=float(buf[14]))
            red = dict(zip("HLA", [float(buf[15]), float(buf[16]), float(buf[17])]), B1=float(buf[18]), B2=float(buf[19]))
            out["device_spans"] = {
                "R1CS to QAP witness map": out["witness_map"],
                "scalar digits + bucket scatter": out["msm_sort"],
                "Compute C": {"h_accumulate": acc["H"], "h_reduce": red["H"], "l_accumulate": acc["L"], "l_reduce": red["L"]},
                "Compute A": {"accumulate": acc["A"], "reduce": red["A"]},
                "Compute B in G1": {"accumulate": acc["B1"], "reduce": red["B1"]},
                "Compute B in G2": {"accumulate": acc["B2"], "reduce": red["B2"]},
                "Finish C": out["host_tail"],
            }
        if n >= 22:     # host Horner over window sums: H's runs after the proof's last device event, the others under H's device work
            out["host_horner_h"], out["host_horner_others"] = float(buf[20]), float(buf[21])
        return out

    def kernel_timing(self, enable=True):
        """0/False off, 1/True all kernel families, 2 only the bucket accumulations"""
        self._check(self.lib.zkg16_kernel_timing(self.ctx, int(enable)))

    def kernel_stats(self, name):
        launches, ms, units = C.c_uint64(), C.c_double(), C.c_double()
        self._check(self.lib.zkg16_kernel_stats(self.ctx, name.encode(), C.byref(launches), C.byref(ms), C.byref(units)))
        return dict(launches=launches.value, ms=ms.value, units=units.value)

    def kernel_stats_reset(self):
        self.lib.zkg16_kernel_stats_reset(self.ctx)

    def set_option(self, name, value):
        self._check(self.lib.zkg16_set_option(self.ctx, name.encode(), int(value)))


def z_costs(r1cs, z_mont, num_instance):
    """Per-index cost of the z-side MSM terms in G1 mixed additions, for shard_plan: (entries of the scalar: 0 for a zero, 1 for a
    one, one per window otherwise) x (queries whose base exists: A if the variable occurs in A or is an instance variable, L if it
    is a witness, B1 + 2.8 x B2 if it occurs in B — ark-groth16 keeps the point at infinity for the rest)."""
    z = _u64(z_mont).reshape(-1, 4)
    m = z.shape[0]
    n = m + 3
    nwin = 254 // (17 if n >= (1 << 23) else 16 if n >= (1 << 20) else 15 if n >= (1 << 17) else 13 if n >= (1 << 14) else max(4, n.bit_length() - 4)) + 1
    one = np.array([0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f], dtype=np.uint64)      # 2^256 mod r
    is_zero = ~z.any(axis=1)
    is_one = (z == one).all(axis=1)
    entries = np.where(is_zero, 0.0, np.where(is_one, 1.0, float(nwin)))
    in_a = np.zeros(m, dtype=bool)
    in_a[np.asarray(r1cs["a"][1], dtype=np.int64)] = True
    in_a[:num_instance] = True
    in_b = np.zeros(m, dtype=bool)
    in_b[np.asarray(r1cs["b"][1], dtype=np.int64)] = True
    mult = in_a.astype(np.float32) + (np.arange(m) >= num_instance) + in_b * 3.8
    return (entries * mult).astype(np.float32)


def shard_plan(n_ranks, m_total, n_h, b_density=0.0, h_ranks=0, z_cost=None, window_tables=False):
    """Rank roles of one proof over n_ranks GPUs (host-only zkg16_shard_plan / zkg16_shard_plan_tables) ->
    (list of (z_lo, z_hi, h_lo, h_hi, blinding) per rank, number of ranks that run the witness map).
    z_cost: optional per-index costs (z_costs) so that the z ranges are cut by work, not by index count.
    window_tables: the shards will carry window tables (pk_precompute): the cost factors measured for that case."""
    lib = _lib.load()
    ranges = np.zeros(4 * n_ranks, dtype=np.uint64)
    blind = np.zeros(n_ranks, dtype=np.uint8)
    k = C.c_int(0)
    zc = None if z_cost is None else np.ascontiguousarray(z_cost, dtype=np.float32)
    if zc is not None and zc.shape[0] != m_total:
        raise ValueError("shard_plan: z_cost has %d entries, m_total is %d" % (zc.shape[0], m_total))
    rc = lib.zkg16_shard_plan_tables(n_ranks, m_total, n_h, float(b_density), h_ranks, _ptr(zc), int(bool(window_tables)), ranges, blind, C.byref(k))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    r = ranges.reshape(n_ranks, 4)
    return [(int(r[i, 0]), int(r[i, 1]), int(r[i, 2]), int(r[i, 3]), bool(blind[i])) for i in range(n_ranks)], k.value


def combine_partials(alpha_g1, beta_g1, beta_g2, r, s, partials, partial_inf):
    """Host-only finish step (no GPU): see zkg16_combine_partials in include/zkg16.h."""
    lib = _lib.load()
    partials = _u64(partials).reshape(-1, 72)
    partial_inf = np.ascontiguousarray(partial_inf, dtype=np.uint8).reshape(-1, 5)
    proof = np.zeros(48, dtype=np.uint64)
    inf = np.zeros(3, dtype=np.uint8)
    rc = lib.zkg16_combine_partials(_u64(alpha_g1), _u64(beta_g1), _u64(beta_g2), _u64(r), _u64(s), partials, partial_inf,
                                    partials.shape[0], proof, inf)
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return proof, inf


def _setup(self, r1cs_h, num_instance, num_vars, domain, trapdoor_mont, g1_gen, g2_gen):
    """Groth16 setup on the device from a known trapdoor (zkg16_setup) -> (pk dict for pk_load, vk dict)."""
    nw = num_vars - num_instance
    pk = dict(a_query=np.zeros((num_vars, 12), np.uint64), a_inf=np.zeros(num_vars, np.uint8),
              b_g1_query=np.zeros((num_vars, 12), np.uint64), b_g1_inf=np.zeros(num_vars, np.uint8),
              b_g2_query=np.zeros((num_vars, 24), np.uint64), b_g2_inf=np.zeros(num_vars, np.uint8),
              h_query=np.zeros((max(domain - 1, 1), 12), np.uint64), l_query=np.zeros((max(nw, 1), 12), np.uint64),
              l_inf=np.zeros(max(nw, 1), np.uint8),
              alpha_g1=np.zeros(12, np.uint64), beta_g1=np.zeros(12, np.uint64), beta_g2=np.zeros(24, np.uint64),
              delta_g1=np.zeros(12, np.uint64), delta_g2=np.zeros(24, np.uint64))
    vk = dict(gamma_g2=np.zeros(24, np.uint64), gamma_abc_g1=np.zeros((num_instance, 12), np.uint64))
    self._check(self.lib.zkg16_setup(
        self.ctx, r1cs_h, _u64(trapdoor_mont).reshape(-1), _u64(g1_gen), _u64(g2_gen),
        pk["a_query"], _ptr(pk["a_inf"]), pk["b_g1_query"], _ptr(pk["b_g1_inf"]), pk["b_g2_query"], _ptr(pk["b_g2_inf"]),
        _ptr(pk["h_query"]), _ptr(pk["l_query"]), _ptr(pk["l_inf"]),
        pk["alpha_g1"], pk["beta_g1"], pk["beta_g2"], pk["delta_g1"], pk["delta_g2"], vk["gamma_g2"], vk["gamma_abc_g1"]))
    pk["h_query"] = pk["h_query"][:domain - 1]
    pk["l_query"], pk["l_inf"] = pk["l_query"][:nw], pk["l_inf"][:nw]
    vk.update(alpha_g1=pk["alpha_g1"], beta_g2=pk["beta_g2"], delta_g2=pk["delta_g2"])
    return pk, vk


Device.setup = _setup


def verify(vk, public_inputs_mont, proof48, inf3):
    """Host-only Groth16 verification (zkg16_verify).  vk: dict alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (n x 12)."""
    lib = _lib.load()
    gabc = _u64(vk["gamma_abc_g1"]).reshape(-1, 12)
    pub = _u64(public_inputs_mont).reshape(-1, 4)
    if pub.shape[0] != gabc.shape[0] - 1:          # a real exception: the C side reads num_instance - 1 inputs
        raise ValueError("verify: %d public inputs for a key with %d instance variables" % (pub.shape[0], gabc.shape[0]))
    ok = C.c_int(0)
    rc = lib.zkg16_verify(_u64(vk["alpha_g1"]), _u64(vk["beta_g2"]), _u64(vk["gamma_g2"]), _u64(vk["delta_g2"]), gabc, gabc.shape[0],
                          _ptr(pub) if pub.size else None, _u64(proof48), np.ascontiguousarray(inf3, dtype=np.uint8), C.byref(ok))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return bool(ok.value)


def _setup_resident(self, r1cs_h, num_instance, trapdoor_mont, g1_gen, g2_gen):
    """Groth16 setup with the key kept on the device (zkg16_setup_resident) -> (pk handle, vk dict)."""
    vk = dict(alpha_g1=np.zeros(12, np.uint64), beta_g2=np.zeros(24, np.uint64), gamma_g2=np.zeros(24, np.uint64),
              delta_g2=np.zeros(24, np.uint64), gamma_abc_g1=np.zeros((num_instance, 12), np.uint64))
    h = C.c_uint64()
    self._check(self.lib.zkg16_setup_resident(self.ctx, r1cs_h, _u64(trapdoor_mont).reshape(-1), _u64(g1_gen), _u64(g2_gen), C.byref(h),
                                              vk["alpha_g1"], vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"], vk["gamma_abc_g1"]))
    return h.value, vk


Device.setup_resident = _setup_resident


def pairing_check(g1_points, g2_points, g1_inf=None, g2_inf=None, plain_final_exp=False):
    """prod e(P_i, Q_i) == 1 on the host (zkg16_pairing_check).  g1_points: n x 12, g2_points: n x 24 Montgomery limbs."""
    lib = _lib.load()
    g1 = _u64(g1_points).reshape(-1, 12)
    g2 = _u64(g2_points).reshape(-1, 24)
    if g1.shape[0] != g2.shape[0]:
        raise ValueError("pairing_check: %d G1 points against %d G2 points" % (g1.shape[0], g2.shape[0]))
    n = g1.shape[0]
    i1 = np.ascontiguousarray(g1_inf if g1_inf is not None else np.zeros(n), dtype=np.uint8)
    i2 = np.ascontiguousarray(g2_inf if g2_inf is not None else np.zeros(n), dtype=np.uint8)
    ok = C.c_int(0)
    rc = lib.zkg16_pairing_check(g1, i1, g2, i2, n, 1 if plain_final_exp else 0, C.byref(ok))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return bool(ok.value)


def scalar_mul(group, base, k_canonical):
    """[k] base on the host (zkg16_scalar_mul_g1/g2) -> (affine limbs, inf)."""
    lib = _lib.load()
    w = 12 if group == "g1" else 24
    out = np.zeros(w, dtype=np.uint64)
    inf = C.c_uint8(0)
    fn = lib.zkg16_scalar_mul_g1 if group == "g1" else lib.zkg16_scalar_mul_g2
    rc = fn(_u64(base).reshape(-1), _u64(k_canonical).reshape(-1), out, C.byref(inf))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return out, int(inf.value)


def point_check(group, point):
    """Curve + prime-order-subgroup membership of one affine point (zkg16_point_check; host-only)."""
    lib = _lib.load()
    ok = C.c_int(0)
    rc = lib.zkg16_point_check(1 if group == "g1" else 2, _u64(point).reshape(-1), C.byref(ok))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return bool(ok.value)


def pvk_prepare(vk):
    """prepare_verifying_key (host-only zkg16_pvk_prepare): vk dict -> the same dict plus alpha_beta (72 u64: ark's Fq12 tower
    order) and gamma_neg_pc / delta_neg_pc (68 x 36 u64 line coefficients each)."""
    lib = _lib.load()
    ab = np.zeros(72, dtype=np.uint64)
    g = np.zeros(68 * 36, dtype=np.uint64)
    d = np.zeros(68 * 36, dtype=np.uint64)
    n = C.c_size_t(0)
    rc = lib.zkg16_pvk_prepare(_u64(vk["alpha_g1"]), _u64(vk["beta_g2"]), _u64(vk["gamma_g2"]), _u64(vk["delta_g2"]), ab, g, d, C.byref(n))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    out = dict(vk)
    out.update(alpha_beta=ab, gamma_neg_pc=g.reshape(n.value, 36), delta_neg_pc=d.reshape(n.value, 36))
    return out


def verify_prepared(pvk, public_inputs_mont, proof48, inf3):
    """Groth16::verify_with_processed_vk on a prepared key (host-only zkg16_verify_prepared)."""
    lib = _lib.load()
    gabc = _u64(pvk["gamma_abc_g1"]).reshape(-1, 12)
    pub = _u64(public_inputs_mont).reshape(-1, 4)
    if pub.shape[0] != gabc.shape[0] - 1:
        raise ValueError("verify_prepared: %d public inputs for a key with %d instance variables" % (pub.shape[0], gabc.shape[0]))
    g, d = _u64(pvk["gamma_neg_pc"]).reshape(-1, 36), _u64(pvk["delta_neg_pc"]).reshape(-1, 36)
    ok = C.c_int(0)
    rc = lib.zkg16_verify_prepared(gabc, gabc.shape[0], _ptr(pub) if pub.size else None, _u64(pvk["alpha_beta"]), g, d, g.shape[0],
                                   _u64(proof48), np.ascontiguousarray(inf3, dtype=np.uint8), C.byref(ok))
    if rc != 0:
        raise Zkg16Error(rc, lib.zkg16_strerror(rc).decode())
    return bool(ok.value)
