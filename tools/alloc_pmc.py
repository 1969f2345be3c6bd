"""Companion of alloc_variance.py for `rocprofv3 --pmc ...`: 8 fresh output allocations, 4 launches of the solve kernel each
(dispatch order = allocation order), HIP-event time of each allocation printed so that counters and speed can be matched."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import _lib, batched, synth  # noqa: E402

ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
base = batched.Plan("2s", cols, bands)
base(); torch.cuda.synchronize()
st = torch.cuda.current_stream()
keep = []
for trial in range(8):
    out = {k: torch.empty_like(v) for k, v in base.out.items()}
    keep.append(out)
    keep.append(torch.empty((trial + 1) * 37 * 2**20 // 8, dtype=torch.float64, device="cuda"))
    p = batched.Plan("2s", cols, bands, out=out, workspace=base.workspace)
    p(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3):
        p(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    print(f"alloc {trial}: {e0.elapsed_time(e1) / 3:.4f} ms", flush=True)
