"""Profile summaries -> the constants bench.py reports (profiles/bench_constants_r2.json), so that `roofline.traffic` and
`alu.peak` come from committed measurements of THIS configuration instead of numbers typed into bench.py:
   python tools/pmc_to_json.py --config n128 --fetch <FETCH_SIZE counter_collection.csv> --write <WRITE_SIZE counter_collection.csv>
                               [--microbench <tools/bin/microbench log>] [--note "..."]
FETCH_SIZE / WRITE_SIZE are per-dispatch KiB (rocprofv3 --pmc, one counter per pass, each with --kernel-trace).  gfx950
correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide contiguous streaming reads — applied to the
NTT row pass (contiguous 16 B/lane loads), not to the bucket accumulation's 112 / 224-byte gathers nor to the column pass'
strided segments (calibrated in round 1 on known byte counts: profiles/rocprofv3_pmc_r1_fetch_write.txt)."""
import argparse, csv, json, os, re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
    return {k: v[1] / max(v[0], 1) * 1024.0 for k, v in acc.items()}, {k: v[0] for k, v in acc.items()}


ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True)
ap.add_argument("--fetch")
ap.add_argument("--write")
ap.add_argument("--microbench")
ap.add_argument("--note", default="")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "bench_constants_r3.json"))
a = ap.parse_args()
d = json.load(open(a.out)) if os.path.exists(a.out) else {}
if a.fetch and a.write:
    f, fc = load(a.fetch, "FETCH_SIZE")
    w, wc = load(a.write, "WRITE_SIZE")

    def pick(table, sub):
        ks = [k for k in table if sub in k]
        return sum(table[k] for k in ks) / max(len(ks), 1) if ks else 0.0
    e = d.setdefault(a.config, {})
    e["msm_accumulate_g1_traffic_bytes"] = pick(f, "msm_accumulate_kernel<zk::FqU") + pick(w, "msm_accumulate_kernel<zk::FqU")
    e["msm_accumulate_g2_traffic_bytes"] = pick(f, "msm_accumulate_kernel<zk::Fq2U") + pick(w, "msm_accumulate_kernel<zk::Fq2U")
    cols = pick(f, "ntt_pass_cols_u") + pick(w, "ntt_pass_cols_u")
    rows = 2.0 * pick(f, "ntt_pass_rows_u") + pick(w, "ntt_pass_rows_u")
    e["ntt_transform_traffic_bytes"] = cols + rows
    e["per_kernel_bytes_per_launch"] = {k: {"fetch": f.get(k, 0.0), "write": w.get(k, 0.0), "launches": fc.get(k, 0)} for k in sorted(set(f) | set(w))
                                        if f.get(k, 0.0) + w.get(k, 0.0) > 1e6}
    e["source"] = {"fetch_csv": os.path.relpath(a.fetch, ROOT), "write_csv": os.path.relpath(a.write, ROOT), "note": a.note}
if a.microbench:
    by_waves, mad = {}, 0.0
    for line in open(a.microbench):
        m = re.match(r"madd G1 \(64thr blk\)\s+blocks/CU=(\d+)\s+[\d.]+ ms\s+([\d.]+) Gop/s", line)
        if m and int(m.group(1)) % 4 == 0:      # b blocks of 64 threads per CU = b / 4 waves per SIMD
            by_waves[str(int(m.group(1)) // 4)] = float(m.group(2))
        m = re.match(r"v_mad_u64_u32\s+blocks/CU=\d+\s+[\d.]+ ms\s+([\d.]+) Gop/s", line)
        if m:
            mad = max(mad, float(m.group(1)))
    alu = d.setdefault("alu", {})
    alu["madd_g1_bare_gadd_per_s_by_waves"] = by_waves          # the kernel's occupancy is read back per launch (zkg16_last_acc_waves)
    alu["madd_g1_bare_gadd_per_s"] = by_waves.get("2", 0.0)
    alu["v_mad_u64_u32_gop_per_s"] = mad                        # best measured issue rate of the whole chip (cycles/instr x held clock)
    alu["mads_per_mixed_addition"] = 3542
    alu["mad_bound_gadd_per_s"] = mad / 3542.0
    alu["source"] = os.path.relpath(a.microbench, ROOT) + " (rows 'madd G1 (64thr blk) blocks/CU=4w' and the best 'v_mad_u64_u32' row)"
json.dump(d, open(a.out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: (v if k == "alu" else {x: y for x, y in v.items() if x != "per_kernel_bytes_per_launch"}) for k, v in d.items()}, indent=1))
