"""Kernel time vs relative placement of the output arrays inside ONE slab (no reallocation between settings), repeated on
several slabs: separates 'relative offset between arrays' from 'where the allocation landed'."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
base = batched.Plan(scheme, cols, bands)
base(); torch.cuda.synchronize()
keys = list(base.out.keys())
shapes = {k: tuple(v.shape) for k, v in base.out.items()}
nbytes = base.out[keys[0]].numel() * 8
st = torch.cuda.current_stream()
MB2 = 2**21


def timeit(plan):
    plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(8):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8


skews = [m * MB2 for m in range(0, 40)] + [m * 65536 for m in (1, 2, 3, 5, 8, 13, 16, 24)] + [2**27, 2**28, 2**29, 2**30 - 1440743424 % 2**30, 2**31 - 1440743424]
keep = []
for s_i in range(3):
    pitch = ((nbytes + MB2 - 1) // MB2) * MB2
    slab = torch.empty(len(keys) * (pitch + max(skews)) + 2 * MB2, dtype=torch.uint8, device="cuda")
    keep.append(slab)
    a0 = (-slab.data_ptr()) % MB2
    res = []
    for skew in skews:
        out = {}
        for i, k in enumerate(keys):
            o = a0 + i * (pitch + skew)
            out[k] = slab[o:o + nbytes].view(torch.float64).view(shapes[k])
        res.append(timeit(batched.Plan(scheme, cols, bands, out=out, workspace=base.workspace)))
    print(f"slab {s_i} (array pitch = {pitch} + x): " + " ".join(f"{sk // 65536}:{t:.3f}" for sk, t in zip(skews, res)), flush=True)
