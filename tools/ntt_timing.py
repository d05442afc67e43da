"""Stand-alone NTT / witness-map timings under library options (development probe):
   python tools/ntt_timing.py [opt=v0,v1 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zksnark_finalproject_amd import Device
dev = Device(0)
sweeps = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in sys.argv[1:]] or [("ntt_xcd", [1])]
circs = {}
for n in (32, 128):
    c, _, _ = bench.synthesize("matrix", n)
    circs[n] = (dev.r1cs_load(c.r1cs, c.num_vars), dev.witness_load(c.z))
for opt, vals in sweeps:
    for v in vals:
        dev.set_option(opt, v)
        line = ["%s=%d:" % (opt, v)]
        for log_n in (16, 19, 20, 22, 24):
            dev.bench_ntt(log_n, 1, 1, 2)
            line.append("2^%d %.3f ms" % (log_n, dev.bench_ntt(log_n, 1, 1, 8)))
        for n, (rh, wh) in circs.items():
            line.append("witness map n=%d %.3f ms" % (n, dev.bench_witness_map(rh, wh, 4)))
        print("  ".join(line), flush=True)
    dev.set_option(opt, 0)
