"""Latency of the drop-in single-canopy path (BASELINE configs[0]: Model(scheme).run() on the default 60 x 107 canopy): host arrays in,
host arrays out, one column per launch.  usage: python tools/model_latency.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd.model import Model

out = {}
for scheme in ("2s", "4s", "bl", "g77", "bf", "n79", "zq", "zq_pa"):
    m = Model(scheme, nlayers=60)
    m.run()  # first call: library load, constant upload
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        m.run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    t1 = time.perf_counter()
    m.calc_absorption()
    ta = time.perf_counter() - t1
    out[scheme] = {"run_ms_median": round(1e3 * ts[len(ts) // 2], 3), "run_ms_min": round(1e3 * ts[0], 3), "calc_absorption_ms": round(1e3 * ta, 3)}
print(json.dumps(out))
