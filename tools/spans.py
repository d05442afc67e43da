"""Device spans of one proof per option value (development probe): python tools/spans.py <matrix_n|prime> <opt> v0,v1,.. [tables=0|1]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device
kind = sys.argv[1]
opt, vals = sys.argv[2], [int(v) for v in sys.argv[3].split(",")]
tables = not (len(sys.argv) > 4 and sys.argv[4] == "tables=0")
dev = Device(0)
trap, g1, g2 = bench.draw_key_inputs(7)
c, _, desc = bench.synthesize("matrix" if kind.isdigit() else kind, int(kind) if kind.isdigit() else 0)
rh = dev.r1cs_load(c.r1cs, c.num_vars)
ph, vk = dev.setup_resident(rh, c.num_instance, trap, g1, g2)
wh = dev.witness_load(c.z)
r, s = bench.fr_mont(12345), bench.fr_mont(67890)
if tables:
    dev.pk_precompute(ph)
print(desc)
for v in vals:
    dev.set_option(opt, v)
    for _ in range(4):
        dev.prove_resident(ph, rh, wh, r, s)
    t = dev.last_timings()
    print(opt, v, "total_wall %.2f host_tail %.3f host_horner_h %.3f others %.3f" % (t["total_wall"], t["host_tail"], t.get("host_horner_h", -1), t.get("host_horner_others", -1)))
    print(json.dumps(t["device_spans"]))
