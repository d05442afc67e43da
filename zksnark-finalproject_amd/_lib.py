"""Loader for the HIP product library libzkg16.so (include/zkg16.h).

There is no CPU fallback anywhere in this package: if the library is missing or no gfx950 device is
visible, importing works (so CPU-only hosts can run the symbol/ABI tests) but creating a context raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKG16_LIB") or os.path.join(HERE, "libzkg16.so")      # ZKG16_LIB: an A/B build of the same ABI (tools/)

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
vp = C.c_void_p
sz = C.c_size_t
H = C.c_uint64
ctxp = C.c_void_p

# name -> (restype, argtypes): every symbol include/zkg16.h declares
SIGNATURES = {
    "zkg16_init": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(ctxp)]),
    "zkg16_destroy": (None, [ctxp]),
    "zkg16_strerror": (C.c_char_p, [C.c_int]),
    "zkg16_last_error": (C.c_char_p, [ctxp]),
    "zkg16_version": (C.c_char_p, []),
    "zkg16_pk_load": (C.c_int, [ctxp, u64p, vp, sz, u64p, vp, sz, u64p, vp, sz, vp, vp, sz, vp, vp, sz,
                                u64p, u64p, u64p, u64p, u64p, sz, C.c_int, C.c_int, C.POINTER(H)]),
    "zkg16_pk_load_range": (C.c_int, [ctxp, u64p, vp, sz, u64p, vp, sz, u64p, vp, sz, vp, vp, sz, vp, vp, sz,
                                      u64p, u64p, u64p, u64p, u64p, sz, sz, sz, sz, sz, C.c_int, C.POINTER(H)]),
    "zkg16_pk_slice": (C.c_int, [ctxp, H, sz, sz, sz, sz, C.c_int, C.POINTER(H)]),
    "zkg16_pk_precompute": (C.c_int, [ctxp, H, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "zkg16_last_term_counts": (C.c_int, [ctxp, C.POINTER(C.c_uint64)]),
    "zkg16_lane_log": (C.c_int, [ctxp, C.POINTER(C.c_double), C.c_int]),
    "zkg16_last_acc_waves": (C.c_int, [ctxp, C.POINTER(C.c_int)]),
    "zkg16_acc_resident_waves": (C.c_int, [ctxp, C.POINTER(C.c_int)]),
    "zkg16_pk_table_bits": (C.c_int, [ctxp, H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "zkg16_shard_plan": (C.c_int, [C.c_int, sz, sz, C.c_double, C.c_int, vp, u64p, u8p, C.POINTER(C.c_int)]),
    "zkg16_shard_plan_tables": (C.c_int, [C.c_int, sz, sz, C.c_double, C.c_int, vp, C.c_int, u64p, u8p, C.POINTER(C.c_int)]),
    "zkg16_pk_free": (None, [ctxp, H]),
    "zkg16_r1cs_load": (C.c_int, [ctxp] + [u64p, vp, vp] * 3 + [sz, sz, sz, C.POINTER(H)]),
    "zkg16_r1cs_free": (None, [ctxp, H]),
    "zkg16_witness_load": (C.c_int, [ctxp, u64p, sz, C.POINTER(H)]),
    "zkg16_witness_free": (None, [ctxp, H]),
    "zkg16_witness_read": (C.c_int, [ctxp, H, u64p, sz]),
    "zkg16_prove_resident": (C.c_int, [ctxp, H, H, H, u64p, u64p, u64p, u8p]),
    "zkg16_prove": (C.c_int, [ctxp, H, u64p, u64p] + [u64p, vp, vp] * 3 + [sz, sz, u64p, sz, u64p, u8p]),
    "zkg16_prove_partial": (C.c_int, [ctxp, H, H, H, u64p, u64p, u64p, u8p]),
    "zkg16_prove_finish": (C.c_int, [ctxp, H, u64p, u64p, u64p, u8p, C.c_int, u64p, u8p]),
    "zkg16_combine_partials": (C.c_int, [u64p, u64p, u64p, u64p, u64p, u64p, u8p, C.c_int, u64p, u8p]),
    "zkg16_setup": (C.c_int, [ctxp, H, u64p, u64p, u64p, u64p, vp, u64p, vp, u64p, vp, vp, vp, vp, u64p, u64p, u64p, u64p, u64p, u64p, u64p]),
    "zkg16_setup_resident": (C.c_int, [ctxp, H, u64p, u64p, u64p, C.POINTER(H), u64p, u64p, u64p, u64p, u64p]),
    "zkg16_scalar_mul_g1": (C.c_int, [u64p, u64p, u64p, C.POINTER(C.c_uint8)]),
    "zkg16_scalar_mul_g2": (C.c_int, [u64p, u64p, u64p, C.POINTER(C.c_uint8)]),
    "zkg16_pairing_check": (C.c_int, [u64p, u8p, u64p, u8p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "zkg16_pvk_prepare": (C.c_int, [u64p, u64p, u64p, u64p, u64p, u64p, u64p, C.POINTER(sz)]),
    "zkg16_verify_prepared": (C.c_int, [u64p, sz, vp, u64p, u64p, u64p, sz, u64p, u8p, C.POINTER(C.c_int)]),
    "zkg16_point_check": (C.c_int, [C.c_int, u64p, C.POINTER(C.c_int)]),
    "zkg16_g1_decompress": (C.c_int, [vp, sz, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "zkg16_g2_decompress": (C.c_int, [vp, sz, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "zkg16_points_compress": (C.c_int, [C.c_int, vp, vp, sz, vp]),
    "zkg16_fq_to_le_bytes": (C.c_int, [vp, sz, vp]),
    "zkg16_fq_from_le_bytes": (C.c_int, [vp, sz, vp]),
    "zkg16_verify": (C.c_int, [u64p, u64p, u64p, u64p, u64p, sz, vp, u64p, u8p, C.POINTER(C.c_int)]),
    "zkg16_circuit_matrix": (C.c_int, [sz, u64p, u64p, C.POINTER(vp)]),
    "zkg16_circuit_matrix_witness": (C.c_int, [sz, u64p, u64p, u64p, sz]),
    "zkg16_circuit_fibonacci": (C.c_int, [C.c_uint64, C.c_uint64, sz, C.POINTER(vp)]),
    "zkg16_prime_search": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), u8p, C.POINTER(C.c_int)]),
    "zkg16_prime_candidate": (C.c_int, [C.c_uint64, C.c_uint64, u8p, C.POINTER(C.c_uint32), u32p, u8p, C.POINTER(C.c_int)]),
    "zkg16_circuit_prime": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "zkg16_prime_public_inputs": (C.c_int, [C.c_uint64, C.c_uint64, u64p]),
    "zkg16_circuit_free": (None, [vp]),
    "zkg16_circuit_dims": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz * 3)]),
    "zkg16_circuit_is_satisfied": (C.c_int, [vp]),
    "zkg16_circuit_export": (C.c_int, [vp, C.POINTER(vp * 3), C.POINTER(vp * 3), C.POINTER(vp * 3), u64p]),
    "zkg16_circuit_public_inputs": (C.c_int, [vp, u64p, sz]),
    "zkg16_circuit_load": (C.c_int, [ctxp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zkg16_poseidon_hash": (C.c_int, [u64p, sz, u64p]),
    "zkg16_r1cs_matrix": (C.c_int, [ctxp, sz, C.POINTER(H)]),
    "zkg16_r1cs_read": (C.c_int, [ctxp, H, vp, vp, vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz * 3)]),
    "zkg16_matrix_r1cs_dims": (C.c_int, [sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz * 3)]),
    "zkg16_matrix_r1cs_host": (C.c_int, [sz, C.POINTER(vp * 3), C.POINTER(vp * 3), C.POINTER(vp * 3)]),
    "zkg16_witness_matrix": (C.c_int, [ctxp, sz, u64p, u64p, C.POINTER(H), vp, vp]),
    "zkg16_prove_matrix": (C.c_int, [ctxp, H, H, sz, u64p, u64p, u64p, u64p, u64p, u8p, vp, vp]),
    "zkg16_matrix_sponge_states": (C.c_int, [sz, u64p, u64p, vp, u64p]),
    "zkg16_ntt": (C.c_int, [ctxp, u64p, sz, C.c_int, C.c_int]),
    "zkg16_msm_g1": (C.c_int, [ctxp, vp, vp, vp, sz, u64p, u8p]),
    "zkg16_msm_g2": (C.c_int, [ctxp, vp, vp, vp, sz, u64p, u8p]),
    "zkg16_witness_map": (C.c_int, [ctxp, H, H, u64p, C.POINTER(sz)]),
    "zkg16_fixed_base_g1": (C.c_int, [ctxp, u64p, vp, sz, vp, vp]),
    "zkg16_fixed_base_g2": (C.c_int, [ctxp, u64p, vp, sz, vp, vp]),
    "zkg16_bench_ntt": (C.c_int, [ctxp, sz, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "zkg16_bench_witness_map": (C.c_int, [ctxp, H, H, C.c_int, C.POINTER(C.c_float)]),
    "zkg16_bench_msm": (C.c_int, [ctxp, C.c_int, vp, vp, vp, sz, C.c_int, C.POINTER(C.c_float), u64p, u8p]),
    "zkg16_last_timings": (C.c_int, [ctxp, C.POINTER(C.c_float), C.c_int]),
    "zkg16_kernel_timing": (C.c_int, [ctxp, C.c_int]),
    "zkg16_kernel_stats": (C.c_int, [ctxp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "zkg16_kernel_stats_reset": (None, [ctxp]),
    "zkg16_set_option": (C.c_int, [ctxp, C.c_char_p, C.c_int64]),
}

_lib = None


class Zkg16Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("zkg16 status %d: %s" % (status, msg))
        self.status = status


def load():
    """dlopen libzkg16.so and attach signatures.  Raises (loudly) if the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libzkg16.so is not built (%s). Run `python zksnark-finalproject_amd/build.py` "
                              "or __graft_entry__.build(); this package has no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
