"""BASELINE.json configs[4] on one GPU: the all-solver sweep (bl / 2s / 4s / gd = g77 / n79 / zq; + bf, zq_pa) on a 1e6 (column x band)
grid, nz = 60, fp64 and f32 storage: kernel time, solves/s, and the observed f32-vs-f64 difference (the only difference allowed is the
final rounding of each output element to float).  One JSON line per scheme."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crt1d_amd import _lib, batched, synth

ncol, nb, nz = 3334, 300, 60  # 1.0002e6 solves
d = synth.make_columns(ncol, nb, nz, seed=5)
d32 = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
d64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in d32.items()}  # the same float-representable inputs in fp64
cols = batched.Columns.from_host(d)
st = torch.cuda.current_stream()


def timed(plan, reps=20):
    plan()
    for _ in range(3):
        plan(flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


for scheme in ("bl", "2s", "4s", "g77", "n79", "zq", "bf", "zq_pa"):
    p64 = batched.Plan(scheme, cols, batched.Bands.from_host(d64))
    p32 = batched.Plan(scheme, cols, batched.Bands.from_host(d32))
    t64 = timed(p64)
    k64 = p64.last_kernel()
    t32 = timed(p32)
    worst = 0.0
    for k in p64.out:
        a, b = p32.out[k].double(), p64.out[k]
        scale = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
        den = torch.maximum(b.abs(), 1e-6 * scale)
        worst = max(worst, float(((a - b).abs() / den).max()))
    print(json.dumps({"config": "BASELINE configs[4] sweep, one GPU", "scheme": scheme, "shape": [ncol, nb, nz], "solves": ncol * nb,
                      "f64": {"kernel_ms": round(t64, 4), "solves_per_s": round(ncol * nb / t64 * 1e3, 0), "kernel": k64},
                      "f32_storage": {"kernel_ms": round(t32, 4), "solves_per_s": round(ncol * nb / t32 * 1e3, 0), "kernel": p32.last_kernel()},
                      "f32_vs_f64_max_elementwise_rel": worst, "bound_2^-24": 2.0 ** -24}), flush=True)
    assert worst <= 2.0 ** -24 * 1.0001, (scheme, worst)
    del p64, p32
