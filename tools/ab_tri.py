"""A/B of the tridiagonal column-tile kernel (M = checkpoint spacing, T = flush tile) vs the per-wave kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "n79"
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ncol, nb = 10000, 300
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"
d = synth.make_columns(ncol, nb, nz, uniform_dlai=not ragged)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan(scheme, cols, bands)
lib = _lib.load()
# (M, T, tune key 10 [1 = k_tri_tile, 2 = double-buffer pipeline, 3 = register-staged pipeline], store waves)
variants = {"tile M12 T12": (12, 12, 1, 0), "tile M16 T8": (16, 8, 1, 0),
            "pipe-db M12 T4 s4": (12, 4, 2, 4), "pipe-db M8 T4 s3": (8, 4, 2, 3), "pipe-db M16 T4 s4": (16, 4, 2, 4),
            "pipe-rs M12 T4 s3": (12, 4, 3, 3), "pipe-rs M12 T4 s4": (12, 4, 3, 4), "pipe-rs M16 T4 s3": (16, 4, 3, 3), "pipe-rs M8 T4 s3": (8, 4, 3, 3),
            "auto": (0, 0, 0, 0), "per-wave": None}
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
for rnd in range(5):
    for name, mt in variants.items():
        flags = _lib.FLAG_SKIP_PRECOMPUTE
        if mt is None:
            flags |= _lib.FLAG_DIRECT_STORES
        else:
            plan.set_tune({8: mt[0], 9: mt[1], 10: mt[2], 11: mt[3]})
        plan(flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=flags)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
for name, v in res.items():
    v = sorted(v)
    print(f"{scheme} nz={nz} ragged={ragged} {name:18s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms")
