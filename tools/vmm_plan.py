"""Real 2s kernel: outputs from torch.empty vs from crt_hip_buffer_alloc (1 GB physical chunks); several candidate sets each, kept alive."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
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
    for _ in range(6):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 6

keep = []
for rnd in range(6):
    a = batched.alloc_outputs(scheme, ncol, nz, nb, "cuda", placed=False)
    b = batched.alloc_outputs(scheme, ncol, nz, nb, "cuda", placed=True)
    keep += [a, b]
    print(f"round {rnd}: torch.empty {timeit(a):.4f} ms   1 GB chunks {timeit(b):.4f} ms", flush=True)
ref = base.out
p = batched.Plan(scheme, cols, bands, out=keep[1], workspace=base.workspace); p(); torch.cuda.synchronize()
print("results equal:", all(torch.equal(p.out[k], ref[k]) for k in ref))
