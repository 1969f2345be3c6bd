"""Timeline of the compute role of the tridiagonal pipelines (k_tri_pipe / k_zqpa_pipe) from a DIAGNOSTIC build
(tools/build_variant.sh stamptri <unit> "-DCRT_STAMP", unit = tri_zqpa | tri_zq_f64 | tri_n79_f64; run with CRT1D_HIP_LIB=...):
compute wave 0 of every workgroup stamps wall_clock64 (100 MHz) into the column's K0 record.  Medians over workgroups, microseconds.

    python tools/stamp_timeline_tri.py scheme ncol nb nz [--ragged]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crt1d_amd import batched, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
scheme = args[0]
ncol, nb, nz = (int(x) for x in args[1:4])
d = synth.make_columns(ncol, nb, nz, seed=1234, uniform_dlai="--ragged" not in sys.argv)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
tune = {int(k): int(v) for k, v in (kv.split(":") for kv in opts["tune"].split(","))} if "tune" in opts else {}
plan = batched.Plan(scheme, cols, bands, tune=tune)
for _ in range(3):
    plan()
torch.cuda.synchronize()
name = plan.last_kernel()
M = int(name.split("M=")[1].split()[0])
T = int(name.split("T=")[1].split()[0])
reclen = (16 + {"n79": 7, "zq": 1, "zq_pa": 4}[scheme] * nz)
ws = plan.workspace.view(torch.float64)[: ncol * reclen].view(ncol, reclen).cpu().numpy()
K = {"n79": nz, "zq": nz + 1, "zq_pa": min(100, nz) + 1}[scheme]
nseg = (K - 1) // M + 1
tick = 0.01
sel = slice(512, ncol - 512)
t = (ws - ws[:, :1]) * tick
med = lambda x: float(np.median(x[sel]))  # noqa: E731
print(f"{name}: K = {K} rows, {nseg} segments of {M}, tiles of {T}; medians over {ncol - 1024} workgroups, us since the record was staged")
print(f"band set-up done {med(t[:, 1]):.2f}; forward sweep done {med(t[:, 2]):.2f}")
if "pipe2" in name:  # zq_pa, interpolation in the compute lanes: the tiles are OUTPUT tiles, not tied to the segments
    i = 3
    print("segment starts / recompute and the output tiles in between (compute, wait at the hand-over):")
    ev = t[:, 3:]
    # reconstruct: per segment 2 stamps, per output tile 2 stamps, in program order; classify by counting tiles = ceil(nz / T)
    ntile = (nz + T - 1) // T
    total = 2 * nseg + 2 * ntile
    m = np.median(ev[sel, :total], axis=0)
    print("  all stamps (us):", " ".join(f"{v:.1f}" for v in m))
    print(f"last stamp {m[-1]:.1f} us; forward {med(t[:, 2] - t[:, 1]):.1f}")
    sys.exit(0)
i = 3
rec_total = back_total = wait_total = 0.0
for seg in range(nseg - 1, -1, -1):
    k0 = seg * M
    kend = min(k0 + M - 1, K - 1)
    ntile = sum(1 for ii in range(M - 1, -1, -1) if k0 + ii <= kend and ii % T == 0)
    s0, s1 = t[:, i], t[:, i + 1]
    i += 2
    line = f"segment {seg}: start {med(s0):7.2f}  pairs recomputed +{med(s1 - s0):5.2f} |"
    rec_total += med(s1 - s0)
    prev = s1
    for _ in range(ntile):
        a, b = t[:, i], t[:, i + 1]
        i += 2
        line += f" tile +{med(a - prev):4.2f} wait {med(b - a):4.2f};"
        back_total += med(a - prev)
        wait_total += med(b - a)
        prev = b
    print(line)
print(f"last hand-over {med(t[:, i - 1]):.2f} us; sums: forward {med(t[:, 2] - t[:, 1]):.1f}, recompute {rec_total:.1f}, back substitution {back_total:.1f}, "
      f"barrier waits {wait_total:.1f}")
