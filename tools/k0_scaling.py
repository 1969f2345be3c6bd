"""K0 (column precompute) time vs number of columns and scheme: fixed launch cost vs per-column cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth

st = torch.cuda.current_stream()
for scheme in ("2s", "zq", "bl"):
    for ncol in (256, 1000, 4096, 10000, 40000, 100000):
        d = synth.make_columns(ncol, 8, 60)
        cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
        plan = batched.Plan(scheme, cols, bands)
        plan(flags=_lib.FLAG_PRECOMPUTE_ONLY); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            plan(st, flags=_lib.FLAG_PRECOMPUTE_ONLY)
        e1.record(st); torch.cuda.synchronize()
        print(f"{scheme} ncol={ncol:6d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
