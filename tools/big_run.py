"""Full-size runs on ONE MI355X: BASELINE config 4 (zq, 1e5 x 300 x 100: 168 GB of outputs) and 2s at 3e5 columns
(5.4e9 elements per output array > 2^32).  Checks: all finite, first/middle/last columns against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crt1d_amd import batched, synth
from oracle import crt_oracle as O

def run(scheme, ncol, nb, nz):
    t0 = time.time()
    d = synth.make_columns(ncol, nb, nz, seed=42)
    print(f"[{scheme}] generated {ncol} x {nb} x {nz} in {time.time()-t0:.1f} s", flush=True)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands)
    nbytes = sum(v.numel() * 8 for v in plan.out.values())
    plan(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"[{scheme}] outputs {nbytes/1e9:.1f} GB, step {ms:.2f} ms -> {ncol*nb/ms*1e3:.3e} solves/s, {nbytes/ms/1e6:.0f} GB/s written", flush=True)
    for k, v in plan.out.items():
        assert bool(torch.isfinite(v).all()), k
    idx = [0, 1, ncol // 2, ncol - 2, ncol - 1]
    oc = O.Columns(d["psi"][idx], d["lai"][idx], mla=d["mla"][idx], g_kind=d["g_kind"][idx], g_param=d["g_param"][idx])
    kw = {k: d[k][idx] for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")}
    ref = O.SOLVERS[scheme](oc, **kw)
    worst = 0.0
    for k, v in plan.out.items():
        got = v[idx].cpu().numpy()
        sc = np.abs(ref[k]).max(axis=1, keepdims=True)
        worst = max(worst, float(np.max(np.abs(got - ref[k]) / sc)))
    print(f"[{scheme}] sampled columns {idx}: max |hip - oracle| / profile max = {worst:.2e}", flush=True)
    assert worst < 1e-9
    del plan, cols, bands
    torch.cuda.empty_cache()

run("zq", 100000, 300, 100)
run("2s", 300000, 300, 60)
print("big runs OK")
