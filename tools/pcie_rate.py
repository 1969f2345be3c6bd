"""PCIe-inclusive rate (SURVEY 8(d) 'GPU timing': H2D + kernel + D2H): what a caller sees when the boundary hands over HOST
buffers.  Pinned host memory, one stream, BASELINE config 2 (2s, 1e4 x 300 x 60).  Never the headline value (bench.py keeps inputs resident)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import batched, synth  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
host_in = {k: torch.from_numpy(d[k]).pin_memory() for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")}
cols = batched.Columns.from_host(d)
dev_in = {k: torch.empty_like(v, device="cuda") for k, v in host_in.items()}
bands = batched.Bands(dev_in["I_dr0"], dev_in["I_df0"], dev_in["leaf_r"], dev_in["leaf_t"], dev_in["soil_r"])
plan = batched.Plan(scheme, cols, bands)
host_out = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in plan.out.items()}


def step():
    for k, v in host_in.items():
        dev_in[k].copy_(v, non_blocking=True)
    out = plan()
    for k, v in out.items():
        host_out[k].copy_(v, non_blocking=True)


step(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = sorted(ts)[len(ts) // 2]
gb_out = sum(v.numel() * v.element_size() for v in host_out.values()) / 1e9
gb_in = sum(v.numel() * v.element_size() for v in host_in.values()) / 1e9
print(f"{scheme} {ncol}x{nb}x{nz}: H2D {gb_in:.2f} GB + solve + D2H {gb_out:.2f} GB = {t * 1e3:.1f} ms per step -> {ncol * nb / t:.3e} solves/s "
      f"({(gb_in + gb_out) / t:.1f} GB/s over PCIe)")
