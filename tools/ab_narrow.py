"""Narrow band shards (the 37/38-band shards of an 8-rank band partition): which kernel family?  One process, interleaved rounds.
usage: python tools/ab_narrow.py scheme ncol nb nz"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1]
ncol, nb, nz = (int(x) for x in sys.argv[2:5])
d = synth.make_columns(ncol, nb, nz, seed=1234)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
tri = scheme in ("n79", "zq")
variants = {"default": {}}
if tri:
    variants.update({"pipeline (auto)": {12: 2}, "k_tri_tile": {12: 2, 10: 1}, "double-buffer pipe": {12: 2, 10: 2}, "regstage pipe": {12: 2, 10: 3},
                     "generic pipe": {12: 2, 10: 4}, "pipe M16 T4": {12: 2, 8: 16, 9: 4}, "pipe M8 T4": {12: 2, 8: 8, 9: 4}, "pipe 2 store waves": {12: 2, 11: 2},
                     "pipe 1 store wave": {12: 2, 11: 1}})
else:
    variants.update({"k_pipe (auto)": {12: 2}, "k_tile CB": {12: 2, 2: 4}, "k_tile CB T=4": {12: 2, 2: 4, 1: 4}, "k_tile CB T=16": {12: 2, 2: 4, 1: 16, 0: 150 * 1024},
                     "k_pipe 1 store wave": {12: 2, 3: 1}, "k_pipe T=8": {12: 2, 4: 8}, "k_pipe T=16": {12: 2, 4: 16}})
plan = batched.Plan(scheme, cols, bands, placement="auto")
plan(); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in plan.out.items()}
st = torch.cuda.current_stream()
res, names = {k: [] for k in variants}, {}
for rnd in range(4):
    for name, tune in variants.items():
        plan.set_tune(tune)
        try:
            plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
        except RuntimeError as e:
            names[name] = f"unsupported ({e})"
            continue
        names[name] = plan.last_kernel()
        if rnd == 0:
            for k in ref:
                assert torch.equal(plan.out[k], ref[k]), (name, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
gb = sum(v.numel() * 8 for v in plan.out.values()) / 1e9
print(f"{scheme} {ncol}x{nb}x{nz} ({gb:.2f} GB written), classes {plan.placement_report and list(plan.placement_report['classes'].values())[:2]}")
for name, v in res.items():
    if v:
        m = sorted(v)[len(v) // 2]
        print(f"  {name:22s} {m:8.3f} ms = {gb / m:5.2f} TB/s   {names[name]}")
    else:
        print(f"  {name:22s} {names.get(name)}")
