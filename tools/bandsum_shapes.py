"""crt_hip_absorb_bandsum_f64 alone over shapes of equal bytes (tools; run on the GPU box): separates per-column overhead from streaming rate.
usage: python tools/bandsum_shapes.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import batched, spectra, synth

st = None
for ncol, nb, nz in ((400000, 38, 25), (200000, 38, 50), (100000, 38, 100), (50000, 38, 200), (25000, 38, 400), (100000, 64, 60), (30000, 64, 200), (100000, 34, 100), (100000, 48, 80), (10000, 300, 60)):
    d = synth.make_columns(ncol, nb, nz, seed=1234)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan("bl", cols, bands)
    sol = plan()
    w = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
    st = torch.cuda.current_stream()
    bs = batched.BandSumPlan(cols, bands, sol, w)
    for _ in range(3):
        bs()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        bs()
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rd = 3 * ncol * nz * nb * 8
    print(json.dumps({"shape": [ncol, nb, nz], "ms": round(ms, 4), "GBs": round(rd / ms / 1e6, 1), "frac_of_8TBs": round(rd / ms / 1e6 / 8000, 3)}), flush=True)
    del bs, plan, sol, cols, bands
