"""Is 'slow' a property of a single 1.44 GB allocation?  All four output pointers of the 2s kernel aimed at the SAME
buffer (timing only; the contents are garbage), for 10 separately allocated buffers; then mixed pairs."""
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
keys = list(base.out.keys())
st = torch.cuda.current_stream()


def timeit(plan):
    plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(8):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8


bufs = []
for i in range(10):
    bufs.append(torch.empty((ncol, nz, nb), dtype=torch.float64, device="cuda"))
    _pad = torch.empty((i + 1) * 29 * 2**20 // 8, dtype=torch.float64, device="cuda")
    bufs.append(_pad)
arrs = bufs[0::2]
single = []
for i, a in enumerate(arrs):
    t = timeit(batched.Plan("2s", cols, bands, out={k: a for k in keys}, workspace=base.workspace))
    single.append(t)
    print(f"buffer {i} (va bits 21.. {(a.data_ptr() >> 21) & 0xfffff:05x}): all four outputs -> it: {t:.4f} ms", flush=True)
order = sorted(range(len(arrs)), key=lambda i: single[i])
fast, slow = order[:4], order[-4:]
for name, idx in (("4 fastest buffers", fast), ("4 slowest buffers", slow), ("2 fastest + 2 slowest", fast[:2] + slow[-2:])):
    t = timeit(batched.Plan("2s", cols, bands, out={k: arrs[i] for k, i in zip(keys, idx)}, workspace=base.workspace))
    print(f"{name} {idx}: {t:.4f} ms", flush=True)
