"""Epilogue kernels alone: crt_hip_absorb_bandsum_f64 (reads 3 profiles) and crt_hip_absorb_f64 (reads 3, writes 7) on the outputs
of a 2s solve.  usage: python tools/epilogue_bench.py [ncol nb nz]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import batched, spectra, synth

ncol, nb, nz = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 300, 60)
d = synth.make_columns(ncol, nb, nz, seed=1234)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan("2s", cols, bands)
sol = plan()
w = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
st = torch.cuda.current_stream()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


bs = batched.BandSumPlan(cols, bands, sol, w)
ms = timeit(bs)
rd = 3 * ncol * nz * nb * 8
print(json.dumps({"kernel": "k_absorb_bandsum", "shape": [ncol, nb, nz], "ms": round(ms, 4), "read_GB": rd / 1e9, "GBs": round(rd / ms / 1e6, 1), "frac_of_8TBs": round(rd / ms / 1e6 / 8000, 3),
                  "classes": plan.placement_report and plan.placement_report["classes"]}))
ms = timeit(lambda: batched.absorb(cols, bands, sol), reps=5)
tr = rd + 7 * ncol * (nz - 1) * nb * 8
print(json.dumps({"kernel": "k_absorb (torch.empty outputs per call)", "shape": [ncol, nb, nz], "ms": round(ms, 4), "bytes_GB": tr / 1e9, "GBs": round(tr / ms / 1e6, 1), "frac_of_8TBs": round(tr / ms / 1e6 / 8000, 3)}))
