#!/usr/bin/env python3
"""bench.py — Groth16 (BLS12-381) proofs/s on the matrix-mul circuit, MI355X HIP path (libzkg16.so).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path = one whole Groth16 proof of the workload (the inside of
`Groth16::<Bls12_381>::prove`: 3 SpMV + 7 NTT + 4 G1 MSM + 1 G2 MSM + tail), with the proving key, the R1CS
matrices and the full assignment already resident in HBM (zkg16_prove_resident).  Workload at N=1 =
BASELINE.json configs[1]: matrix-mul 32x32 + Poseidon circuit shape (472,564 constraints, domain 2^19).
N>1: index-range sharded proving key (every rank recomputes h, runs the five MSMs over its 1/N of the bases),
ONE exchange per proof — an all_gather of 72 u64 of partial sums over RCCL — and the tail on every rank:
strong scaling of one proof, as BASELINE.json's north_star asks (`--parallel replicas` is the throughput alternative:
every rank proves its own proofs on the whole key, no exchange, weak scaling).

At N=1 the line also carries `throughput_in_flight`: after the contract's timed region, the same proofs with
`--in-flight` (default 2) of them in flight at once, one library ctx per host thread — what a multi-worker server gets.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (msm_accumulate_g1) from HIP-event pairs
recorded on the library's stream during the timed region; `cpu_baseline` times the CPU oracle (a port, see
oracle/g16_oracle.c) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--matrix-n", type=int, default=32, help="n of the n x n matrix-mul circuit (32 = configs[1], 46 = 2^20 domain)")
    ap.add_argument("--workload", default="matrix", choices=["matrix", "prime_like"],
                    help="matrix = the reference's MatrixCircuit (metric workload); prime_like = bit-heavy circuit of configs[4]'s shape")
    ap.add_argument("--synthetic-rows", action="store_true", help="shape-exact synthetic rows instead of the synthesized MatrixCircuit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) | gloo (rehearsal: several ranks on one GPU)")
    ap.add_argument("--cpu-sample-n", type=int, default=16, help="matrix size of the bounded CPU-baseline sample")
    ap.add_argument("--parallel", default="shard", choices=["shard", "replicas"],
                    help="N > 1: shard = one proof's index ranges over the ranks (north_star; strong scaling); "
                         "replicas = every rank proves its own proofs on the whole key (no exchange; weak scaling)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="N = 1: after the contract's timed region, also report throughput with this many proofs in flight "
                         "(one ctx per host thread, as one ctx per server worker would run); 0/1 = skip")
    return ap.parse_args()


def rand_fr_mont(rng):
    # any 4 limbs < 2^254 are a valid Montgomery representative of some Fr element
    v = rng.integers(0, 1 << 62, size=4, dtype=np.uint64)
    return v


def make_key(dev, r1cs, shp, seed):
    """A structurally faithful proving key for the workload: every query element is a valid subgroup point
    [k]G with random k (fixed-base kernel on the GPU), and a query entry is the point at infinity exactly where
    arkworks' generator would produce one (variable absent from that side of the R1CS)."""
    from zksnark_finalproject_amd.workloads import g1_generator, g2_generator
    rng = np.random.default_rng(seed)
    nv, ni = shp["num_vars"], shp["num_instance"]
    n_h = shp["domain"] - 1

    def pts(group, n):
        gen = g1_generator() if group == "g1" else g2_generator()
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        p, inf = dev.fixed_base(group, gen, sc)
        return p

    in_a = np.zeros(nv, dtype=bool)
    in_a[r1cs["a"][1]] = True
    in_a[:ni] = True                       # instance rows of the LibsnarkReduction put L_{nc+k} on the A side
    in_b = np.zeros(nv, dtype=bool)
    in_b[r1cs["b"][1]] = True
    pk = {}
    pk["a_query"] = pts("g1", nv)
    pk["a_inf"] = (~in_a).astype(np.uint8)
    pk["b_g1_query"] = pts("g1", nv)
    pk["b_g1_inf"] = (~in_b).astype(np.uint8)
    pk["b_g2_query"] = pts("g2", nv)
    pk["b_g2_inf"] = pk["b_g1_inf"].copy()
    pk["h_query"] = pts("g1", n_h)
    pk["l_query"] = pts("g1", nv - ni)
    single1 = pts("g1", 3)
    single2 = pts("g2", 3)
    pk["alpha_g1"], pk["beta_g1"], pk["delta_g1"] = single1[0], single1[1], single1[2]
    pk["beta_g2"], pk["delta_g2"] = single2[1], single2[2]
    return pk


def cpu_baseline(sample_n, target_nc, threads, dev=None):
    """Times the CPU oracle (port of the arkworks algorithms; oracle/) on a bounded sample of the same workload
    family and scales by constraint count to the metric's unit (proofs/s of the target circuit)."""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import random

    import oracle as orc
    import synth
    from helpers import fr_mont

    from zksnark_finalproject_amd.circuits import matrix_circuit
    circ = matrix_circuit(np.ones((sample_n, sample_n), dtype=np.uint64), np.ones((sample_n, sample_n), dtype=np.uint64))
    r1cs, z = circ.r1cs, circ.z
    shp = dict(nc=circ.num_constraints, num_vars=circ.num_vars, domain=circ.domain)
    rng = random.Random(7)
    # the sample's key: points from the device's fixed-base kernel when a device is at hand (untimed; the timed part is the
    # CPU prover alone)
    pk, _ = synth.make_pk(orc, r1cs, shp["num_vars"], rng, point_gen=dev.fixed_base if dev is not None else None)
    used = orc.set_threads(threads)
    t0 = time.time()
    orc.prove(pk, fr_mont(12345), fr_mont(67890), r1cs, z)
    dt = time.time() - t0
    cps = shp["nc"] / dt
    # the same prover on one thread (the published curve looks single-threaded), on a quarter-size sample
    small = matrix_circuit(np.ones((sample_n // 2, sample_n // 2), dtype=np.uint64), np.ones((sample_n // 2, sample_n // 2), dtype=np.uint64))
    pk1, _ = synth.make_pk(orc, small.r1cs, small.num_vars, rng, point_gen=dev.fixed_base if dev is not None else None)
    orc.set_threads(1)
    t1 = time.time()
    orc.prove(pk1, fr_mont(12345), fr_mont(67890), small.r1cs, small.z)
    dt1 = time.time() - t1
    orc.set_threads(threads)
    return dict(value=cps / target_nc, unit="proofs/s", cores=used, kind="port",
                one_thread={"value": small.num_constraints / dt1 / target_nc, "unit": "proofs/s", "cores": 1,
                            "sample": "n=%d (%d constraints) in %.2f s" % (sample_n // 2, small.num_constraints, dt1)},
                sample="oracle prove of the same MatrixCircuit at n=%d (%d constraints, domain 2^%d) in %.2f s on %d threads "
                       "(OpenMP: one task per MSM window, as ark's `parallel` feature); scaled by constraint count to n=32-equivalent proofs/s"
                       % (sample_n, shp["nc"], shp["domain"].bit_length() - 1, dt, used),
                constraints_per_sec=cps)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    on_gpu = args.dist_backend == "nccl"
    dev_index = local_rank if (world > 1 and on_gpu) else int(os.environ.get("ZKG16_BENCH_DEVICE", "0"))
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    from zksnark_finalproject_amd import Device
    from zksnark_finalproject_amd.workloads import matmul_like_r1cs

    dev = Device(dev_index)
    if args.workload == "prime_like":
        from zksnark_finalproject_amd.workloads import prime_like_r1cs
        r1cs, z, shp = prime_like_r1cs()
    elif args.synthetic_rows:
        r1cs, z, shp = matmul_like_r1cs(args.matrix_n)      # same seed on every rank
    else:
        # the reference's MatrixCircuit itself (C++ mirror, csrc/circuits.hip) on bench/matrix.py:11's all-ones inputs
        from zksnark_finalproject_amd.circuits import matrix_circuit
        n = args.matrix_n
        circ = matrix_circuit(np.ones((n, n), dtype=np.uint64), np.ones((n, n), dtype=np.uint64))
        r1cs, z = circ.r1cs, circ.z
        shp = dict(n=n, nc=circ.num_constraints, num_instance=circ.num_instance, num_witness=circ.num_witness,
                   num_vars=circ.num_vars, domain=circ.domain)
    sharded = world > 1 and args.parallel == "shard"
    pk = make_key(dev, r1cs, shp, seed=0xC0FFEE)
    ph = dev.pk_load(pk, shp["num_instance"], shard_index=rank if sharded else 0, shard_count=world if sharded else 1)
    rh = dev.r1cs_load(r1cs, shp["num_vars"])
    wh = dev.witness_load(z)
    if not (world == 1 and args.in_flight > 1):
        del pk                       # kept for the proofs-in-flight leg's further contexts (created after the timed region)
    rng = np.random.default_rng(99)
    rs = [(rand_fr_mont(rng), rand_fr_mont(rng)) for _ in range(args.steps + args.warmup)]

    xdev = "cuda" if on_gpu else "cpu"
    if sharded:
        gather_buf = [torch.empty(77, dtype=torch.int64, device=xdev) for _ in range(world)]

    def one_proof(r, s):
        if not sharded:
            return dev.prove_resident(ph, rh, wh, r, s)
        part, pinf = dev.prove_partial(ph, rh, wh, r, s)
        rec = np.concatenate([part.view(np.int64), pinf.astype(np.int64)])
        dist.all_gather(gather_buf, torch.from_numpy(rec).to(xdev))        # the single exchange: 77 words per rank over xGMI
        allrec = torch.stack(gather_buf).cpu().numpy()
        parts = allrec[:, :72].copy().view(np.uint64)
        pinfs = allrec[:, 72:].astype(np.uint8)
        return dev.prove_finish(ph, r, s, parts, pinfs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        proof, inf = one_proof(*rs[i])
    dev.kernel_stats_reset()
    dev.kernel_timing(2)             # async HIP-event pairs around the bucket accumulations (the roofline kernel) on the
                                     # library's own streams; resolved after the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        proof, inf = one_proof(*rs[args.warmup + i])
    barrier()
    dt = time.perf_counter() - t0
    dev.kernel_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stages = dev.last_timings()
    in_flight = None
    extra = []                       # further contexts on the same GPU, made only now so that the contract's region saw one ctx
    if world == 1 and args.in_flight > 1:
        for _ in range(args.in_flight - 1):
            d2 = Device(dev_index)
            extra.append((d2, d2.pk_load(pk, shp["num_instance"]), d2.r1cs_load(r1cs, shp["num_vars"]), d2.witness_load(z)))
        del pk
    if extra:
        import threading
        lanes = [(dev, ph, rh, wh)] + extra
        per = max(args.steps, 4)

        def lane_work(lane, count):
            d, p_, r_, w_ = lane
            for j in range(count):
                d.prove_resident(p_, r_, w_, *rs[j % len(rs)])
        for lane in lanes[1:]:
            lane_work(lane, 1)
        ths = [threading.Thread(target=lane_work, args=(lane, per)) for lane in lanes]
        barrier()
        t1 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        barrier()
        dt2 = time.perf_counter() - t1
        in_flight = {"proofs_in_flight": len(lanes), "proofs": per * len(lanes), "value": per * len(lanes) / dt2, "unit": "proofs/s",
                     "ms_per_proof": dt2 / (per * len(lanes)) * 1e3,
                     "note": "same GPU, one library ctx (own streams + workspaces) per host thread; outside the contract's timed region"}
        for d2, *_ in extra:
            d2.close()

    if rank == 0:
        total_proofs = args.steps * (world if (world > 1 and not sharded) else 1)
        acc = dev.kernel_stats("msm_accumulate_g1")
        acc2 = dev.kernel_stats("msm_accumulate_g2")
        # algorithmic bytes of one G1 bucket-accumulation launch: every (base, scalar) term once = (96 + 32) B per term
        # (SURVEY.md 8d); terms per launch = MSM length of that launch, averaged over the 4 G1 MSMs of a proof.
        nshard = world if sharded else 1
        nz = (shp["num_vars"] + 3 + nshard - 1) // nshard
        nh = (shp["domain"] - 1 + nshard - 1) // nshard
        terms_per_launch = (3 * nz + nh) / 4.0
        alg_bytes = 128.0 * terms_per_launch
        avg_ms = acc["ms"] / max(acc["launches"], 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic of one msm_accumulate_g1 launch from rocprofv3 PMC passes of this same command (profiles/
        # rocprofv3_pmc_r1_fetch_write.txt: FETCH_SIZE + WRITE_SIZE, calibrated on ntt_pass_cols: no 2x for 112-B gathers);
        # only known for the default workload on one GPU
        traffic = 1.05e9 if (world == 1 and args.matrix_n == 32 and args.workload == "matrix") else None
        out = {
            "metric": "groth16_proofs_per_sec", "value": total_proofs / dt, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (sharded or world == 1) else "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": ("Fermat-prime-shaped boolean circuit (BASELINE configs[4] shape, synthetic rows): %d constraints, %d witness vars, domain 2^%d; "
                                    "pk/R1CS/assignment resident in HBM" % (shp["nc"], shp["num_witness"], shp["domain"].bit_length() - 1))
                       if args.workload == "prime_like" else
                                   "matrix-mul %dx%d + Poseidon circuit (BASELINE configs[1] when n=32): %d constraints, %d witness vars, domain 2^%d; "
                                   "pk/R1CS/assignment resident in HBM; %s; structurally faithful random-point key"
                                   % (args.matrix_n, args.matrix_n, shp["nc"], shp["num_witness"], shp["domain"].bit_length() - 1,
                                      "synthetic shape-exact rows" if args.synthetic_rows else "R1CS + witness synthesized by the C++ mirror of the reference's MatrixCircuit on all-ones inputs"),
                       "parallelism": "1 GPU" if world == 1 else
                                      ("index-range sharded pk over %d GPUs + 1 all_gather(77 words)/proof" % world if sharded else
                                       "%d replicas: every GPU proves its own proofs on the whole key, no exchange" % world)},
            "constraints_per_sec": shp["nc"] * total_proofs / dt,
            "stage_ms_last_proof": stages,
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_g1", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": acc["launches"], "algorithmic_bytes_per_launch": alg_bytes,
                         "terms_per_launch": acc["units"] / max(acc["launches"], 1),
                         "note": "integer-ALU bound by construction: one XYZZ mixed addition = 8 products + 2 squarings in Fq (~4,700 VALU "
                                 "instructions) per window per 128 algorithmic bytes; traffic = PMC FETCH+WRITE of the same command "
                                 "(bases are re-read once per window); g2 accumulate avg %.3f ms" % (acc2["ms"] / max(acc2["launches"], 1))},
        }
        # the bound that does apply: mixed additions per second against the same addition in a bare register-resident loop
        # (tools/microbench.hip, profiles/microbench_r1_uform.txt: 6.75 G add/s at 2 waves/SIMD on this part)
        tpl = acc["units"] / max(acc["launches"], 1)
        nwin = 254 // (16 if tpl >= (1 << 20) else 15 if tpl >= (1 << 17) else 13 if tpl >= (1 << 14) else 9) + 1
        # entries per launch: a scalar 1 gives one entry, a scalar 0 none, everything else one per window (L, A, B1 share
        # z; H is uniform); averaged over the four G1 launches like avg_ms
        R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
        one = np.array([((1 << 256) % R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        zz = np.asarray(z, dtype=np.uint64).reshape(-1, 4)
        n_one = int(np.all(zz == one, axis=1).sum())
        n_zero = int(np.all(zz == 0, axis=1).sum())
        z_entries = ((zz.shape[0] - n_one - n_zero) * nwin + n_one) / nshard
        h_entries = nh * (254 // (16 if nh >= (1 << 20) else 15 if nh >= (1 << 17) else 13 if nh >= (1 << 14) else 9) + 1)
        gadd = (3 * z_entries + h_entries) / 4.0 / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out["alu"] = {"kernel": "msm_accumulate_g1", "achieved": gadd, "peak": 6.75, "unit": "G mixed additions/s", "frac": gadd / 6.75,
                      "note": "entries = one per window for every scalar other than 0 and 1 (B1's density filter not counted: slight "
                              "over-estimate); peak = the XYZZ mixed addition alone in a register-resident loop at the kernel's occupancy"}
        # second kernel family of the path: one transform of the workload's domain, timed alone after the timed region
        # (64 algorithmic bytes per element: each element read once and written once)
        try:
            log_n = shp["domain"].bit_length() - 1
            dev.bench_ntt(log_n, 1, 1, 2)
            ntt_ms = dev.bench_ntt(log_n, 1, 1, 10)
            out["roofline_ntt"] = {"bound": "hbm", "kernel": "ntt_pass_cols_u + ntt_pass_rows_u (one coset-inverse transform of 2^%d)" % log_n,
                                   "achieved": 64.0 * shp["domain"] / (ntt_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": 64.0 * shp["domain"] / (ntt_ms * 1e-3) / 1e9 / 8000.0, "ms": ntt_ms, "transforms_per_proof": 7,
                                   "note": "integer-ALU bound too: ~7 Fr products per element per pass, two passes"}
        except Exception as e:      # noqa: BLE001 - the extra leg must never cost the contract line
            out["roofline_ntt"] = {"error": repr(e)}
        if in_flight:
            out["throughput_in_flight"] = in_flight
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, shp["nc"], min(os.cpu_count() or 1, 16), dev)
        print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
