"""zq_pa at odd nb: two-kernel path vs the single-kernel form with the flat store role (tools; run on the GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
for shape in ((30000, 107, 60), (10000, 299, 60), (100000, 37, 100), (6000, 301, 100)):
    d = synth.make_columns(*shape)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan("zq_pa", cols, bands)
    plan(); torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    res = {}
    for name, tune in {"two kernels": {10: 1}, "fused s1": {11: 1}, "fused s2": {11: 2}, "fused s3": {11: 3}, "fused s4": {11: 4}, "default": {}}.items():
        plan.set_tune(tune)
        plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
        k = plan.last_kernel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(st); torch.cuda.synchronize()
        res[name] = (round(e0.elapsed_time(e1) / 5, 3), k[:44])
    print(shape, res, flush=True)
    del plan
