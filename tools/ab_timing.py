"""Is the T=8 tile kernel 1.03 ms or 1.16 ms?  Same process: back-to-back timing vs per-launch events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

d = synth.make_columns(10000, 300, 60)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan("2s", cols, bands)
lib = _lib.load()
st = torch.cuda.current_stream()
F = _lib.FLAG_SKIP_PRECOMPUTE
def b2b(n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): plan(st, flags=F)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def per_launch(n=20):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in evs:
        e0.record(st); plan(st, flags=F); e1.record(st)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(t) / n, t[0], t[n // 2]
plan(); torch.cuda.synchronize()
for name, tune in (("auto", {0: 78 * 1024, 1: 0}), ("forced T=8", {0: 160 * 1024, 1: 8}), ("forced T=4", {0: 40960, 1: 4}), ("auto again", {0: 78 * 1024, 1: 0})):
    plan.set_tune(tune)
    plan(flags=F); torch.cuda.synchronize()
    print(f"{name:12s} back-to-back {b2b():.4f} ms | per-launch avg/min/med {per_launch()} | back-to-back {b2b():.4f}")
