"""Per-kernel averages of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output):
   python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import csv, re, sys
from collections import defaultdict
def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        acc[k][0] += 1
        acc[k][1] += float(row["Counter_Value"])
    return acc
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("%-62s %6s %16s %16s %14s" % ("kernel", "calls", "FETCH_SIZE/call", "WRITE_SIZE/call", "(F+W) MB/call"))
rows = []
for k in f:
    fc = f[k][1] / max(f[k][0], 1)
    wc = w[k][1] / max(w[k][0], 1) if k in w else 0.0
    rows.append((f[k][1] + (w[k][1] if k in w else 0), k, f[k][0], fc, wc))
for _, k, calls, fc, wc in sorted(rows, reverse=True)[:24]:
    print("%-62s %6d %16.1f %16.1f %14.1f" % (k[:62], calls, fc, wc, (fc + wc) * 1024 / 1e6))
