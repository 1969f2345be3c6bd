"""Candidate output sets whose arrays are each preceded by a random-size pad allocation (pads freed after the set exists):
how often is such a set 'fast'?  argv: scheme nsets seed free_pads(0/1)"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 8
random.seed(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
free_pads = len(sys.argv) > 4 and sys.argv[4] == "1"
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


keep, res = [], [timeit(base.out)]
for i in range(nsets):
    out, pads = {}, []
    for k, v in base.out.items():
        pads.append(torch.empty(random.randrange(1, 150) * 2**21, dtype=torch.uint8, device="cuda"))
        out[k] = torch.empty_like(v)
    if free_pads:
        del pads
        torch.cuda.empty_cache()
    else:
        keep.append(pads)
    keep.append(out)
    res.append(timeit(out))
print(f"{scheme} free_pads={free_pads}: back-to-back {res[0]:.3f} | padded sets: " + " ".join(f"{t:.3f}" for t in res[1:]), flush=True)
