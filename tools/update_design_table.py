"""Regenerate the measurement table of DESIGN.md section 5 from profiles/r03/rocprof/summary.json (rows between the headline row and the
'(Before the streaming stores' note).  usage: python tools/update_design_table.py"""
import json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(ROOT, "profiles/r03/rocprof/summary.json")))
desc = [
    ("2s", "2s 1e4×300×60 (**headline**)"), ("2s_125k", "2s 125000×300×60 (north-star per-GPU share, 73.5 GB of outputs)"), ("bl", "bl"), ("4s", "4s"), ("g77", "g77"), ("bf", "bf"),
    ("n79", "n79 (plain stores: two of its arrays have nz−1 rows)"), ("n79_nz100", "n79 nz=100 (6000×300×100)"), ("zq", "zq"), ("zq_nz100", "zq nz=100 (6000×300×100)"),
    ("band_cfg4", "zq 25000×300×100 (one column tile of the band-mode bench, config 4)"), ("zq_nb38_nz100", "zq **1e5×38×100** (a rank's shard of config 4)"),
    ("2s_nb38", "2s 2e5×38×60"), ("2s_nb107", "2s 3e4×107×60 (the reference's 107 bands)"), ("n79_nb107", "n79 3e4×107×60"), ("zq_nb107", "zq 3e4×107×60"), ("zq_pa", "zq_pa"),
    ("zq_pa_nz100", "zq_pa nz=100 (6000×300×100)"), ("zq_pa_nb107", "zq_pa 3e4×107×60 (odd nb: flat store role)"), ("zq_pa_nb38", "zq_pa 1e5×38×100"), ("2s_f32", "2s, f32 storage"),
    ("n79_f32", "n79, f32 storage"), ("zq_nb12", "zq 4e5×12×60 (packed: 5 columns per compute wave)"), ("n79_nb12", "n79 4e5×12×60 (packed)"), ("2s_nb12", "2s 4e5×12×60 (packed)"),
    ("4s_nb12", "4s 4e5×12×60 (packed)"), ("zq_nb8_wave", "zq 4e5×8×60 (packed: 8 columns per compute wave; `k_tri_wave` in round 2)"),
    ("2s_ragged", "**ragged ΔLAI** 2s"), ("bl_ragged", "ragged bl"), ("4s_ragged", "ragged 4s"), ("g77_ragged", "ragged g77"), ("bf_ragged", "ragged bf"), ("n79_ragged", "ragged n79"),
    ("n79_nz100_ragged", "ragged n79 nz=100 (6000×300×100)"), ("zq_ragged", "ragged zq"), ("zq_nz100_ragged", "ragged zq nz=100"), ("zq_pa_ragged", "ragged zq_pa"),
]
rows = []
# the same bench commands WITHOUT the profiler (tools/bench_cases_noprof.py): HIP-event kernel time -> fraction of the peak
npp = os.path.join(ROOT, "profiles/r03/bench/noprof_summary.json")
noprof = json.load(open(npp)) if os.path.exists(npp) else {}
for k, d in desc:
    if k not in s:
        continue
    e = s[k]
    kern = e["bench"]["kernel"].split(" lds=")[0]
    ev = noprof.get(k, {}).get("frac")
    evs = f" ({ev:.2f})" if ev else ""
    rows.append(f"| {d} | `{kern}` | {e['rocprof_avg_ms']:.3f} | {e['rocprof_GBs'] / 1e3:.2f} | {e['rocprof_frac_of_8TBs']:.2f}{evs} | {e.get('traffic_over_algorithmic', float('nan')):.3f} |")


def ms(d, sub):
    for k, v in d.items():
        if sub in k:
            return v["avg_us"] / 1e3


def hb(d, sub):
    for k, v in d.items():
        if sub in k:
            return v["hbm_bytes_per_launch"]


ep, ep38, bz = s["epilogue"], s["epilogue_nb38"], s["band_zq"]
t = ms(ep["kernels"], "bandsum")
rows.append(f"| epilogue: band sums of 3 profiles, 1e4×300×60 | `k_absorb_bandsum_w<5,2,3>` | {t:.3f} | {4.32 / t:.2f} (read) | {4.32 / t / 8:.2f} | {hb(ep['pmc'], 'bandsum') / 4.32e9:.2f} |")
t = ms(ep38["kernels"], "bandsum")
rows.append(f"| epilogue: band sums, 1e5×38×100 (a rank's shard) | `k_absorb_bandsum_l<3>` | {t:.3f} | {9.12 / t:.2f} (read) | {9.12 / t / 8:.2f} | {hb(ep38['pmc'], 'bandsum') / 9.12e9:.2f} (17 rows fetched per 16 layers + 0.7 GB written) |")
t = ms(ep["kernels"], "absorb_tile")
rows.append(f"| epilogue: 7 per-band outputs | `k_absorb_tile` | {t:.3f} | {14.232 / t:.2f} (read + write) | {14.232 / t / 8:.2f} | {hb(ep['pmc'], 'absorb_tile') / 14.232e9:.3f} |")
t1, t2 = ms(bz["kernels"], "k_tri_pipe"), ms(bz["kernels"], "bandsum")
rows.append(f"| band-partition step, one tile 5000×300×100 (`band_zq`) | `k_tri_pipe<zq,…>` + `k_absorb_bandsum_w` | {t1:.3f} + {t2:.3f} | {8.46 / t1:.2f} / {3.6 / t2:.2f} | {8.46 / t1 / 8:.2f} / {3.6 / t2 / 8:.2f} | — |")
i = [s[k]["rocprof_avg_ms"] for k in ("2s_integrated", "n79_integrated", "zq_integrated")]
rows.append(f"| integrated outputs only (2s / n79 / zq) | `k_int`, `k_tri_int` (wave_sum4 reductions) | {i[0]:.3f} / {i[1]:.3f} / {i[2]:.3f} | — | VALU-bound | — |")
p = os.path.join(ROOT, "DESIGN.md")
d = open(p).read()
t0 = d.index("| 2s 1e4×300×60 (**headline**) |")
t1 = d.index("\n(rocprof averages include the warm-up launches")
open(p, "w").write(d[:t0] + "\n".join(rows) + "\n" + d[t1:])
print("\n".join(rows))
