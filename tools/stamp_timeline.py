"""Per-workgroup timeline of k_pipe from a DIAGNOSTIC build (tools/build_variant.sh stamp solve_closed "-DCRT_STAMP"; run with
CRT1D_HIP_LIB=variants/libcrt1d_hip_stamp.so): compute wave 0 and store wave 0 of every workgroup stamp wall_clock64 (100 MHz) into
the column's own K0 record in the workspace.  Prints medians over the workgroups, in microseconds since the workgroup's first stamp.

    python tools/stamp_timeline.py [scheme ncol nb nz] [--ragged]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crt1d_amd import batched, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
scheme = args[0] if args else "4s"
ncol, nb, nz = (int(x) for x in args[1:4]) if len(args) >= 4 else (10000, 300, 60)
ragged = "--ragged" in sys.argv
d = synth.make_columns(ncol, nb, nz, seed=1234, uniform_dlai=not ragged)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan(scheme, cols, bands)
for _ in range(3):
    plan()
torch.cuda.synchronize()
reclen = batched.workspace_bytes(scheme, ncol, nz, nb) // 8 // ncol
ws = plan.workspace.view(torch.float64)[: ncol * reclen].view(ncol, reclen).cpu().numpy()
T = int(plan.last_kernel().split("T=")[1].split()[0])
ntile = (nz + T - 1) // T
comp = ws[:, : 2 + 2 * ntile]
store = ws[:, 64 : 64 + 2 + 2 * ntile]
tick = 0.01  # us per wall_clock64 tick (100 MHz)
t0 = np.minimum(comp[:, 0], store[:, 0])
c = (comp - t0[:, None]) * tick
s = (store - t0[:, None]) * tick
sel = slice(512, ncol - 512)  # steady state: skip the first and last wave of workgroups
med = lambda x: float(np.median(x[sel]))  # noqa: E731
print(f"{plan.last_kernel()}  ({'ragged' if ragged else 'uniform'} dlai), medians over {ncol - 1024} workgroups, us")
print(f"compute wave 0: init done {med(c[:, 1]):.2f}")
print("  tile  compute-done  barrier-passed | store: barrier-passed  stores-issued")
for g in range(ntile):
    print(f"  {g:3d}   {med(c[:, 2 + 2 * g]):8.2f}     {med(c[:, 3 + 2 * g]):8.2f}      |        {med(s[:, 1 + 2 * g]):8.2f}       {med(s[:, 2 + 2 * g]):8.2f}")
print(f"all stores acknowledged {med(s[:, 1 + 2 * ntile]):.2f}")
cd = np.diff(np.concatenate([c[:, 1:2], c[:, 2::2]], axis=1), axis=1)  # per tile: previous barrier-pass ... hmm: compute-done minus previous barrier-passed
comp_time = c[:, 2 + 2 * np.arange(ntile)] - np.concatenate([c[:, 1:2], c[:, 3 + 2 * np.arange(ntile - 1)]], axis=1)
wait_time = c[:, 3 + 2 * np.arange(ntile)] - c[:, 2 + 2 * np.arange(ntile)]
issue_time = s[:, 2 + 2 * np.arange(ntile)] - s[:, 1 + 2 * np.arange(ntile)]
swait = s[:, 1 + 2 * np.arange(ntile)] - np.concatenate([s[:, 0:1], s[:, 2 + 2 * np.arange(ntile - 1)]], axis=1)
print(f"per tile (median over tiles 2.. and workgroups): compute {np.median(comp_time[sel, 2:]):.2f} us, compute wave waits at the barrier {np.median(wait_time[sel, 2:]):.2f} us; "
      f"store wave: issue {np.median(issue_time[sel, 2:]):.2f} us, waits at the barrier {np.median(swait[sel, 2:]):.2f} us")
print(f"sums per column (median): compute {np.median(comp_time[sel].sum(axis=1)):.1f}, compute-wave barrier wait {np.median(wait_time[sel].sum(axis=1)):.1f}, "
      f"store issue {np.median(issue_time[sel].sum(axis=1)):.1f}, store-wave barrier wait {np.median(swait[sel].sum(axis=1)):.1f}, column life {med(s[:, 1 + 2 * ntile]):.1f}")

dbg = ws[:, 100:100 + 3 * 8].reshape(ncol, 8, 3)
print("HW_ID / LDS_ALLOC / virtual wave of the 8 waves, four sample workgroups (hex):")
for c_ in (600, 601, 5000, 5001):
    print("  wg", c_, " ".join(f"{int(dbg[c_, w, 0]):08x}/{int(dbg[c_, w, 1]):08x}/{int(dbg[c_, w, 2])}" for w in range(8)))
hw = dbg[:, :, 0].astype(np.int64)
for lo, width, name in ((0, 4, "wave_id[3:0]"), (4, 2, "simd[5:4]"), (6, 2, "pipe[7:6]"), (8, 4, "cu[11:8]"), (12, 1, "sh"), (13, 3, "se[15:13]")):
    f = (hw >> lo) & ((1 << width) - 1)
    print(f"  {name}: values seen {sorted(set(f.flatten().tolist()))[:20]}; distinct per workgroup (median) {np.median([len(set(r)) for r in f[sel]])}")
simd = (hw >> 4) & 3
vw = dbg[:, :, 2].astype(np.int64)
ncw = 5
cc = np.array([[((simd[c_] == sd) & (vw[c_] < ncw)).sum() for sd in range(4)] for c_ in range(512, 1536)])
print("compute waves per SIMD within a workgroup: pattern counts", {k: int(v) for k, v in zip(*np.unique(np.sort(cc, axis=1), axis=0, return_counts=True))} if False else "", np.unique(np.sort(cc, axis=1), axis=0, return_counts=True))
