"""Integrated-output variant (reported separately from the headline): fused kernel vs solve + epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import batched, spectra, synth

ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
w = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
st = torch.cuda.current_stream()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for scheme in ("2s", "4s", "bl", "g77", "n79", "zq"):
    ip = batched.IntegratedPlan(scheme, cols, bands, w)
    sp = batched.Plan(scheme, cols, bands)
    t_f = timeit(lambda: ip())
    t_u = timeit(lambda: (sp(), batched.absorb_bandsum(cols, bands, sp.out, w)))
    print(f"{scheme:4s} fused {t_f:.3f} ms = {ncol*nb/t_f*1e3:.3e} solves/s | solve + epilogue {t_u:.3f} ms = {ncol*nb/t_u*1e3:.3e} solves/s")
