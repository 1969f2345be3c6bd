"""Output sets whose arrays are separated by multi-GB pad allocations (pads freed once the set exists): does physical
distance between the arrays of a set make the fast mode the rule?  argv: pad_gb list"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth

scheme = "2s"
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
base = batched.Plan(scheme, cols, bands)
base(); torch.cuda.synchronize()
st = torch.cuda.current_stream()

def timeit(out):
    plan = batched.Plan(scheme, cols, bands, out=out, workspace=base.workspace)
    plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(4):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4

keep = []
for pad_gb in (0, 1, 2, 4, 8, 16, 3, 5, 7):
    res = []
    for trial in range(4):
        out, pads = {}, []
        for k, v in base.out.items():
            if pad_gb:
                pads.append(torch.empty(pad_gb << 30, dtype=torch.uint8, device="cuda"))
            out[k] = torch.empty_like(v)
        del pads
        torch.cuda.empty_cache()
        keep.append(out)
        res.append(timeit(out))
    print(f"pads of {pad_gb:2d} GB between the arrays of a set: " + " ".join(f"{t:.3f}" for t in res), flush=True)
    if len(keep) > 20:
        keep = keep[-8:]
        torch.cuda.empty_cache()
