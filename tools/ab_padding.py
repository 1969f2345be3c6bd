"""Does the relative placement of the 4 output arrays matter (HBM channel aliasing)?  Same process, same kernels,
outputs carved from one big buffer with different gaps between arrays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
lib = _lib.load()
st = torch.cuda.current_stream()
F = _lib.FLAG_SKIP_PRECOMPUTE
n = ncol * nz * nb
big = torch.empty(4 * n + 4 * (1 << 24), dtype=torch.float64, device="cuda")
ws = torch.empty(batched.workspace_bytes("2s", ncol, nz), dtype=torch.uint8, device="cuda")
def run(gap_doubles, flags, reps=3):
    outs = {}
    off = 0
    for k in ("I_dr", "I_df_d", "I_df_u", "F"):
        outs[k] = big[off:off + n].view(ncol, nz, nb)
        off += n + gap_doubles
    plan = batched.Plan("2s", cols, bands, out=outs, workspace=ws)
    plan(); torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): plan(st, flags=flags)
        e1.record(st); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10)
    return min(best)
print("base ptr %x" % big.data_ptr())
for gap in (0, 16, 512, 4096 // 8, 65536 // 8, (1 << 20) // 8, (1 << 21) // 8 + 16, 3 * (1 << 20) // 8 + 4096 // 8, 12345 * 16, (1 << 24) - 16):
    t_tile = run(gap, F)
    t_dir = run(gap, F | _lib.FLAG_DIRECT_STORES)
    print(f"gap {gap*8:>10d} B: tile(T=8) {t_tile:.4f} ms   direct {t_dir:.4f} ms")
# separately allocated (torch default) for reference
plan = batched.Plan("2s", cols, bands)
plan(); torch.cuda.synchronize()
for flags, nm in ((F, "tile"), (F | _lib.FLAG_DIRECT_STORES, "direct")):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): plan(st, flags=flags)
    e1.record(st); torch.cuda.synchronize()
    print("torch-allocated outputs", nm, e0.elapsed_time(e1) / 10, [hex(v.data_ptr()) for v in plan.out.values()])
