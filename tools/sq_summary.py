"""Per-kernel wave statistics from rocprofv3 --pmc SQ_* passes (csv counter_collection files; several passes may be given —
every pass must carry SQ_WAVES and SQ_WAVE_CYCLES, its other counters are normalised by ITS OWN wave cycles, then the passes are
merged by kernel name):  python tools/sq_summary.py <counter_collection.csv> [...]
Fractions of SQ_WAVE_CYCLES: active = issuing any instruction, wait_any = parked at s_waitcnt / barrier, wait_inst = ready but
not issued.  SQ cycle counters are in quad-cycles; the ratios are unit-free.  VALU/wave = SQ_INSTS_VALU / SQ_WAVES."""
import csv, re, sys
from collections import defaultdict
merged = defaultdict(dict)
for path in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES":
            launches[k] += 1
    for k, c in acc.items():
        wc, waves = c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_WAVES", 0.0)
        if wc <= 0 or waves <= 0:
            continue
        m = merged[k]
        m["wave_cycles"] = max(m.get("wave_cycles", 0.0), wc)
        m["launches"] = launches[k]
        m["waves_per_launch"] = waves / max(launches[k], 1)
        for name, key, denom in (("SQ_INSTS_VALU", "valu_per_wave", waves), ("SQ_INSTS_VMEM", "vmem_per_wave", waves),
                                 ("SQ_ACTIVE_INST_ANY", "active", wc), ("SQ_WAIT_ANY", "wait_any", wc), ("SQ_WAIT_INST_ANY", "wait_inst", wc),
                                 ("SQ_ACTIVE_INST_VALU", "valu_active", wc)):
            if name in c:
                m[key] = c[name] / denom
for k, m in sorted(merged.items(), key=lambda kv: -kv[1]["wave_cycles"])[:28]:
    print("  %-58s launches %4d waves/launch %8d  VALU/wave %10d  VMEM/wave %7d  active %.2f wait_any %.2f wait_inst %.2f valu_active %.2f"
          % (k[:58], m["launches"], m["waves_per_launch"], m.get("valu_per_wave", 0), m.get("vmem_per_wave", 0), m.get("active", 0),
             m.get("wait_any", 0), m.get("wait_inst", 0), m.get("valu_active", 0)))
