"""Builds libzkg16.so (the HIP product library) in-tree for gfx950.

    python zksnark-finalproject_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libzkg16.so")
SOURCES = ["api.hip", "ntt.hip", "poly.hip", "msm.hip", "sort.hip", "bucket_sort.hip", "setup.hip", "circuits.hip", "verify.hip", "witness.hip", "matrix_r1cs.hip"]
HEADERS = ["ff.cuh", "ffu.cuh", "fru.cuh", "ec.cuh", "common.hpp", "hostff.hpp", "matrix_plan.hpp", "pairing_fast.inc", "poseidon_h64.inc", "poseidon_params.inc", "prime_circuit.inc", "final_exp.inc", os.path.join("..", "..", "include", "zkg16.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = ["hipcc"] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
