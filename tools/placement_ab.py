"""Output placement A/B in one process: torch.empty outputs ("none") vs the class-interleaving set allocator ("auto"), solve kernel only
(K0 skipped), 30 launches each, three fresh plans of each kind.  usage: python tools/placement_ab.py [scheme ncol nb nz]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
ncol, nb, nz = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (10000, 300, 60)
d = synth.make_columns(ncol, nb, nz, seed=1234)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
st = torch.cuda.current_stream()


def time_plan(p, reps=30):
    p()
    for _ in range(3):
        p(flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        p(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


keep = []
for rnd in range(3):
    for placement in ("none", "auto"):
        t0 = time.perf_counter()
        p = batched.Plan(scheme, cols, bands, placement=placement)
        torch.cuda.synchronize()
        t_alloc = time.perf_counter() - t0
        ms = time_plan(p)
        print(json.dumps({"scheme": scheme, "shape": [ncol, nb, nz], "round": rnd, "placement": placement, "kernel_ms": round(ms, 4),
                          "plan_s": round(t_alloc, 3), "classes": None if p.placement_report is None else p.placement_report["classes"]}), flush=True)
        keep.append(p)  # held: the next plan lands elsewhere
print(json.dumps(batched.buffer_stats()))
