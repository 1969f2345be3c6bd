"""(The flush-order / nontemporal variants this tool once compared -- profiles/r01/alloc_variance.txt -- made no difference and
were removed from the kernel.)  Does the solve kernel's speed depend on WHERE its output arrays live?  One process, the same kernel and inputs, output
arrays re-allocated several ways: fresh torch allocations (previous ones kept alive), and slices of one slab with a skew
of k * skew_bytes between consecutive arrays."""
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
st = torch.cuda.current_stream()


def timeit(plan):
    plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(st); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    return sorted(ts)[1]


keep = []
print("fresh allocations (address bits 21..30 of each array in hex):")
for trial in range(6):
    out = {k: torch.empty(shapes[k], dtype=torch.float64, device="cuda") for k in keys}
    keep.append(out)
    keep.append(torch.empty((trial + 1) * 37 * 2**20 // 8, dtype=torch.float64, device="cuda"))  # perturb the next addresses
    p = batched.Plan(scheme, cols, bands, out=out, workspace=base.workspace)
    lib = _lib.load()
    t = timeit(p)
    print(f"  trial {trial}: {t:.4f} ms   " + " ".join(f"{(v.data_ptr() >> 21) & 0x3ff:03x}" for v in out.values()), flush=True)
