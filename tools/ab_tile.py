"""A/B of tile-kernel variants in ONE process, interleaved rounds (guide rule 24). Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
if len(sys.argv) > 2 and sys.argv[2] == "f32":
    import numpy as np
    d = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan(scheme, cols, bands, placement="auto")
lib = _lib.load()
# tune keys: 0 = LDS target bytes, 1 = force T (k_tile), 2 = flags (bit1 generic flush, bit2 no pipeline), 3 = store waves, 4 = pipeline T
variants = {
    "k_pipe default": {0: 78 * 1024, 1: 0, 2: 0, 3: 0, 4: 0},
    "T=4 s3": {0: 78 * 1024, 1: 0, 2: 0, 3: 3, 4: 4},
    "T=4 s2": {0: 78 * 1024, 1: 0, 2: 0, 3: 2, 4: 4},
    "T=8 s2": {0: 78 * 1024, 1: 0, 2: 0, 3: 2, 4: 8},
    "T=8 s4": {0: 78 * 1024, 1: 0, 2: 0, 3: 4, 4: 8},
    "T=16 s3": {0: 78 * 1024, 1: 0, 2: 0, 3: 3, 4: 16},
    "k_tile T=8": {0: 78 * 1024, 1: 8, 2: 4, 3: 0, 4: 0},
    "k_tile T=16": {0: 78 * 1024, 1: 16, 2: 4, 3: 0, 4: 0},
}
res = {k: [] for k in variants}
plan(); torch.cuda.synchronize()  # K0 once: the timed launches below skip the precompute and reuse this workspace
buf = torch.empty(2 * 10**9 // 8, dtype=torch.float64, device="cuda")
def fill_rate():
    st_ = torch.cuda.current_stream().cuda_stream
    lib.crt_hip_probe_fill_f64(buf.data_ptr(), buf.numel(), 1.0, st_); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.crt_hip_probe_fill_f64(buf.data_ptr(), buf.numel(), 1.0, st_)
    b.record(); torch.cuda.synchronize()
    return buf.numel() * 8 * 5 / (a.elapsed_time(b) * 1e-3) / 1e9
print(f"fill probe {fill_rate():.0f} GB/s")
st = torch.cuda.current_stream()
for rnd in range(6):
    for name, tune in variants.items():
        flags = _lib.FLAG_SKIP_PRECOMPUTE
        if tune is None:
            flags |= _lib.FLAG_DIRECT_STORES
        else:
            plan.set_tune(tune)
        plan(flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            plan(st, flags=flags)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10)
for name, v in res.items():
    v = sorted(v)
    print(f"{scheme} {name:24s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms")
