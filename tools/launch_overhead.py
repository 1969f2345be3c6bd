"""Host cost of one plan() call (ctypes + 2 launches), measured on a tiny problem so that the GPU is never the limit."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth

for scheme, ncol, nb, nz in [("2s", 64, 300, 60), ("2s", 10000, 300, 60), ("n79", 64, 300, 60)]:
    d = synth.make_columns(ncol, nb, nz)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands)
    for _ in range(10):
        plan()
    torch.cuda.synchronize()
    for flags, name in ((0, "K0 + solve"), (_lib.FLAG_SKIP_PRECOMPUTE, "solve only"), (_lib.FLAG_PRECOMPUTE_ONLY, "K0 only")):
        t0 = time.perf_counter()
        for _ in range(200):
            plan(flags=flags)
        t_enq = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        t_tot = (time.perf_counter() - t0) / 200
        print(f"{scheme} {ncol}x{nb}x{nz} {name:11s}: enqueue {t_enq * 1e6:7.1f} us per call, with completion {t_tot * 1e6:7.1f} us", flush=True)
