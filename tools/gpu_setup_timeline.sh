# kernel timeline of the second setup of a loop: bash tools/gpu_setup_timeline.sh <matrix_n> [min_ms]
R=$GRAFT_REPO_ROOT
N=${1:-128}
MINMS=${2:-0.8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/setup_tl -o tl -- python3 $R/tools/setup_loop.py $N 3 > $R/gpurun_out/setup_tl.log 2>&1
cd $R
MINMS=$MINMS python3 - <<'PY'
import csv, glob, os
f = glob.glob('gpurun_out/setup_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'setup_expand_kernel' in r['Kernel_Name']]
# three expand kernels per setup: the last setup's first expand
i0 = starts[-3]
# back up to the fr_powers kernel before it
while i0 > 0 and 'fr_powers_kernel' not in rows[i0]['Kernel_Name']: i0 -= 1
t0 = int(rows[i0]['Start_Timestamp'])
minms = float(os.environ.get('MINMS', '0.8'))
busy = 0; last_end = t0; gaps = 0
end_idx = None
for k, r in enumerate(rows[i0:]):
    if 'msm_digits_kernel' in r['Kernel_Name']: end_idx = i0 + k; break
sel = rows[i0:end_idx]
print("kernels in this setup:", len(sel), " span %.2f ms" % ((int(sel[-1]['End_Timestamp']) - t0) / 1e6))
cur_end = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > cur_end: gaps += s - cur_end
    cur_end = max(cur_end, e)
print("idle between kernels (no kernel running): %.2f ms" % (gaps / 1e6))
for r in sel:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    if e - s > minms: print("%8.2f -> %8.2f (%6.2f ms) %s" % (s, e, e - s, r['Kernel_Name'][:80]))
PY
rm -rf gpurun_out/setup_tl
