"""Exploratory: every scheme on extreme shapes vs the oracle; prints status / max relative error per shape.
Run on the GPU box: python tools/robust_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crt1d_amd import batched, synth  # noqa: E402
from oracle import crt_oracle as O  # noqa: E402

SHAPES = [(1, 1, 2), (3, 1, 3), (2, 5, 2), (1, 2151, 60), (2, 1025, 20), (1, 4000, 8), (2, 40, 400), (1, 16, 1000), (1, 8, 3000),
          (70000, 1, 5), (1, 300, 60), (3, 1023, 61),
          # round 2: the narrow-band pipelines (16 <= nb < 64) and the flat flush (odd nb) at awkward depths
          (3, 38, 400), (2, 17, 1000), (5, 63, 150), (4, 33, 512), (3, 20, 700), (7, 37, 2), (6, 16, 3), (2, 107, 300), (3, 255, 130), (2, 62, 61)]
SCHEMES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["2s", "4s", "bl", "g77", "bf", "n79", "zq", "zq_pa"]


def main():
    dev = torch.device("cuda:0")
    for shape in SHAPES:
        ncol, nb, nz = shape
        for uniform in (True, False):
            d = synth.make_columns(ncol, nb, nz, seed=7, uniform_dlai=uniform)
            cols = batched.Columns.from_host(d)
            bands = batched.Bands.from_host(d)
            nref = min(ncol, 3)
            oc = O.Columns(d["psi"][:nref], d["lai"][:nref], mla=d["mla"][:nref], g_kind=d["g_kind"][:nref], g_param=d["g_param"][:nref])
            for sch in SCHEMES:
                if sch == "n79" and nz < 3:
                    continue
                t0 = time.time()
                try:
                    sol = batched.solve(sch, cols, bands)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    print(f"{shape} unif={uniform} {sch}: EXC {type(e).__name__}: {e}", flush=True)
                    continue
                kw = dict(I_dr0=d["I_dr0"][:nref], I_df0=d["I_df0"][:nref], leaf_r=d["leaf_r"][:nref], leaf_t=d["leaf_t"][:nref],
                          soil_r=d["soil_r"][:nref])
                if sch == "bl":
                    kw.pop("soil_r")
                if nz * nb > 200000 and sch in ("4s",):
                    ref = None
                else:
                    try:
                        ref = O.SOLVERS[sch](oc, **kw)
                    except Exception as e:  # noqa: BLE001
                        print(f"{shape} unif={uniform} {sch}: oracle EXC {type(e).__name__}: {e}", flush=True)
                        ref = None
                worst = 0.0
                fin = all(bool(torch.isfinite(v).all()) for v in sol.values())
                if ref is not None:
                    for k, v in sol.items():
                        a = v[:nref].cpu().numpy()
                        r = ref[k]
                        scale = np.abs(r).max() + 1e-300
                        worst = max(worst, float(np.abs(a - r).max() / scale))
                print(f"{shape} unif={uniform} {sch}: ok finite={fin} err={worst:.2e} ({time.time() - t0:.2f}s)", flush=True)
                if sch == "2s" and ref is not None:  # the epilogue kernels on the same shape (all their paths: half wave / wave / slices; tile / rows)
                    from crt1d_amd import spectra

                    w = spectra.band_weights(d["wle"])
                    bs = batched.absorb_bandsum(cols, bands, sol, torch.as_tensor(w).cuda())
                    per = batched.absorb(cols, bands, sol)
                    out3 = {k: sol[k][:nref].cpu().numpy() for k in ("I_dr", "I_df_d", "I_df_u")}
                    ab = O.calc_absorption(oc, out3, leaf_r=d["leaf_r"][:nref], leaf_t=d["leaf_t"][:nref])
                    e1 = max(float(np.abs(per[k][:nref].cpu().numpy() - ab[k]).max() / (np.abs(ab[k]).max() + 1e-300)) for k in batched.ABSORPTION_KEYS)
                    e2 = max(float(np.abs(bs[k][:nref].cpu().numpy() - ab[k] @ w.T).max() / (np.abs(ab[k] @ w.T).max() + 1e-300)) for k in ("aI", "aI_sl", "aI_sh"))
                    print(f"{shape} unif={uniform} epilogue: absorb err={e1:.2e} bandsum err={e2:.2e}", flush=True)


if __name__ == "__main__":
    main()
