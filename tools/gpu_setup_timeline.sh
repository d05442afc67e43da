R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/setup_tl -o tl -- python3 $R/tools/setup_loop.py 128 2 > $R/gpurun_out/setup_tl.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/setup_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# second request: find the last setup_expand/fr_powers start
starts = [i for i, r in enumerate(rows) if 'fr_powers_kernel' in r['Kernel_Name']]
# requests: fr_powers appears at start of setup (Lsrc) -> take the first of the last group
t0 = None
grp = []
for i in starts:
    if not grp or int(rows[i]['Start_Timestamp']) - int(rows[grp[-1]]['Start_Timestamp']) < 150e6: grp.append(i)
    else: grp = [i]
i0 = grp[0]
t0 = int(rows[i0]['Start_Timestamp'])
out = []
for r in rows[i0:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    if e - s > 0.8: out.append((s, e, r['Kernel_Name'][:70], r.get('Stream_Id', r.get('Queue_Id', ''))))
for s, e, n, q in out[:70]:
    print("%8.2f -> %8.2f (%6.2f ms) q%s %s" % (s, e, e - s, q, n))
PY
rm -rf gpurun_out/setup_tl
