"""Per-kernel timeline of the last proof in a rocprofv3 --kernel-trace rocpd database (development probe):
   python tools/timeline.py gpurun_out/prof_x/tl_results.db [min_ms]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = db.execute("select name,start,end,queue_id,grid_x,workgroup_x,vgpr_count from kernels order by start").fetchall()
def short(n):
    n = re.sub(r'\(.*', '', n).replace('zk::', '').replace('void ', '')
    n = re.sub(r'rocprim::.*trampoline_kernel<rocprim::ROCPRIM_400200_NS::detail::', 'rp::', n)
    return n[:48]
# a proof starts with the z-side digit kernel: the last proof = from the third-last digit launch on small/medium circuits
# (z, z masked by the B density, h); pass the number of digit launches per proof as argv[3] when it differs
per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [i for i, r in enumerate(rows) if 'msm_digits' in r[0]]
sub = rows[idx[-per]:]
t0 = sub[0][1]
for r in sub:
    if (r[2] - r[1]) / 1e6 >= min_ms:
        print("%8.3f %8.3f  q%-3s grid %-8d vgpr %-4d %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4], r[6], short(r[0])))
print("proof span %.3f ms" % ((max(r[2] for r in sub) - t0) / 1e6))
