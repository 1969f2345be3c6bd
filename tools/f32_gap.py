import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth
d = synth.make_columns(10000, 300, 60)
d32 = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
for name, dd in (("f64", d), ("f32", d32)):
    cols, bands = batched.Columns.from_host(dd), batched.Bands.from_host(dd)
    plan = batched.Plan("2s", cols, bands, placement="auto")
    st = torch.cuda.current_stream()
    print(name, plan.placement_report)
    for flags, what in ((0, "K0+solve"), (_lib.FLAG_SKIP_PRECOMPUTE, "solve only")):
        for _ in range(5): plan(flags=flags)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): plan(flags=flags)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in e:
            a.record(st); plan(st, flags=flags); b.record(st)
        torch.cuda.synchronize()
        ev = sorted(a.elapsed_time(b) for a, b in e)
        print(f"{name} {what:10s}: back-to-back wall {wall*1e3:.4f} ms per call; per-call events median {ev[15]:.4f} ms (min {ev[0]:.4f}, max {ev[-1]:.4f})", flush=True)
