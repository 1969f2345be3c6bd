"""A/B of the fused zq_pa kernel: store waves, vs the two-kernel path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth
lib = _lib.load()
st = torch.cuda.current_stream()
for ncol, nb, nz in [(10000, 300, 60), (6000, 300, 100), (30000, 128, 60)]:
    d = synth.make_columns(ncol, nb, nz)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan("zq_pa", cols, bands, placement="auto")
    plan(); torch.cuda.synchronize()
    variants = {"two kernels": (1, 0), "fused s2": (0, 2), "fused s3": (0, 3), "fused s4": (0, 4), "fused s5": (0, 5)}
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name, (k10, k11) in variants.items():
            plan.set_tune({10: k10, 11: k11})
            plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 5)
    gb = sum(v.numel() * 8 for v in plan.out.values()) / 1e9
    print(f"zq_pa {ncol}x{nb}x{nz} ({gb:.2f} GB): " + "  ".join(f"{k} {sorted(v)[len(v)//2]:.3f} ms" for k, v in res.items()), flush=True)
    del plan, cols, bands
    torch.cuda.empty_cache()
