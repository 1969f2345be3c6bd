"""Random-shape parity fuzz (tools; GPU box): every scheme on random (ncol, nb, nz), uniform and ragged dLAI, f64 against the oracle and f32
storage against the rounded f64 result, plus the epilogue kernels.  usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crt1d_amd import batched, spectra, synth
from oracle import crt_oracle as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
# bars of tests/test_gpu_parity.py (2s: 1e-6 -- the sigma -> 0 removable singularity amplifies operation-order differences; 4s: 1e-7)
TOL = {"2s": 1e-6, "4s": 1e-7, "bl": 1e-6, "g77": 1e-10, "bf": 1e-10, "n79": 1e-6, "zq": 1e-6, "zq_pa": 1e-6}
F32 = ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")
bad = 0
t0 = time.time()
for case in range(n_cases):
    nb = int(rng.choice([rng.integers(1, 20), rng.integers(20, 70), rng.integers(70, 140), rng.integers(140, 420), rng.choice([37, 38, 64, 107, 128, 256, 300, 512, 513])]))
    nz = int(rng.choice([rng.integers(2, 12), rng.integers(12, 70), rng.integers(70, 130), rng.choice([60, 61, 100, 101, 129, 257])]))
    ncol = int(rng.integers(1, 24))
    unif = bool(rng.integers(0, 2))
    d = synth.make_columns(ncol, nb, nz, seed=int(rng.integers(1, 1 << 30)), uniform_dlai=unif)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    b32 = batched.Bands.from_host({k: (d[k].astype(np.float32) if k in F32 else d[k]) for k in d})
    b64r = batched.Bands(*[None if t is None else t.double() for t in (b32.I_dr0, b32.I_df0, b32.leaf_r, b32.leaf_t, b32.soil_r)])
    oc = oracle.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    msgs = []
    for scheme in TOL:
        try:
            sol = batched.solve(scheme, cols, bands)
            kw = {k: d[k] for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")}
            if scheme == "bl":
                kw.pop("soil_r")
            if scheme == "4s":
                kw["method"] = "eig"
            ref = oracle.SOLVERS[scheme](oc, **kw)
            for k in ("I_dr", "I_df_d", "I_df_u", "F"):
                got = sol[k].cpu().numpy()
                sc = np.abs(ref[k]).max(axis=1, keepdims=True)
                sc[sc == 0] = 1
                err = np.max(np.abs(got - ref[k]) / sc)
                if not np.isfinite(got).all() or err > TOL[scheme]:
                    msgs.append(f"{scheme} {k} err {err:.2e}")
            try:
                s32 = batched.solve(scheme, cols, b32)
                s64 = batched.solve(scheme, cols, b64r)
                for k in s32:
                    if not torch.equal(s32[k], s64[k].float()):
                        msgs.append(f"{scheme} f32 {k} differs from rounded f64")
            except RuntimeError as e:
                if "not supported" not in str(e):
                    raise
            if scheme == "2s" and nz >= 2:
                w = spectra.band_weights(d["wle"])
                res = batched.absorb_bandsum(cols, bands, sol, torch.as_tensor(w).cuda())
                per = batched.absorb(cols, bands, sol)
                out = {k: sol[k].cpu().numpy() for k in ("I_dr", "I_df_d", "I_df_u")}
                ab = oracle.calc_absorption(oc, out, leaf_r=d["leaf_r"], leaf_t=d["leaf_t"])
                for k in ("aI", "aI_sl", "aI_sh"):
                    r = ab[k] @ w.T
                    e = np.max(np.abs(res[k].cpu().numpy() - r)) / max(np.abs(r).max(), 1e-300)
                    if e > 1e-11:
                        msgs.append(f"bandsum {k} err {e:.2e}")
                    e = np.max(np.abs(per[k].cpu().numpy() - ab[k])) / max(np.abs(ab[k]).max(), 1e-300)
                    if e > 1e-12:
                        msgs.append(f"absorb {k} err {e:.2e}")
        except AssertionError as e:
            if scheme == "n79" and nz < 3 and "reference assertion" in str(e):
                continue  # as the reference: its n79 asserts on fewer than three levels
            msgs.append(f"{scheme} raised AssertionError: {str(e)[:80]}")
        except Exception as e:  # noqa: BLE001
            msgs.append(f"{scheme} raised {type(e).__name__}: {str(e)[:80]}")
    if msgs:
        bad += 1
        print(f"case {case} shape ({ncol}, {nb}, {nz}) unif={unif}: " + "; ".join(msgs), flush=True)
    elif case % 10 == 0:
        print(f"case {case} shape ({ncol}, {nb}, {nz}) unif={unif}: ok  [{time.time() - t0:.0f} s]", flush=True)
print(f"fuzz done: {n_cases} cases, {bad} with findings, {time.time() - t0:.0f} s")
