"""Store rate of the 2s kernel against the size of its output set (tools; GPU box): ncol from 2.5e3 (1.5 GB) to 1.6e5 (94 GB)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
placement = sys.argv[1] if len(sys.argv) > 1 else "auto"
for ncol in (2500, 5000, 10000, 15000, 20000, 30000, 40000, 80000, 160000):
    d = synth.make_columns(ncol, 300, 60)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan("2s", cols, bands, placement=placement)
    plan(); torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    reps = max(3, int(20000 * 10 / ncol))
    for _ in range(2):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = ncol * 300 * 1960 / 1e9
    cl = plan.placement_report["classes"] if plan.placement_report else None
    print(f"ncol {ncol:7d}  outputs {gb:6.1f} GB  {ms:8.4f} ms  {gb / ms:6.3f} TB/s  frac {gb / ms / 8:5.3f}  classes {list(cl.values())[0][:16] if cl else None}", flush=True)
    del plan, cols, bands
    batched.trim_buffers() if hasattr(batched, "trim_buffers") else None
