"""Markdown rows of DESIGN.md section 5 from a profile summary (tools/summarize_profiles.py output).
usage: python tools/design_table.py profiles/r02/rocprof/summary.json"""
import json, sys

s = json.load(open(sys.argv[1]))
print("| case | kernel (as the library reports it) | launch shape | rocprof avg ms | TB/s | frac of 8 TB/s | PMC traffic / algorithmic |")
print("|---|---|---|---|---|---|---|")
for case, e in s.items():
    b = e.get("bench") or {}
    if not b.get("launch_shape") or "rocprof_avg_ms" not in e:
        continue
    shape = "x".join(str(x) for x in b["launch_shape"])
    print(f"| {case} | `{b.get('kernel')}` | {shape} | {e['rocprof_avg_ms']:.3f} | {e['rocprof_GBs'] / 1e3:.2f} | {e['rocprof_frac_of_8TBs']:.2f} | "
          f"{e.get('traffic_over_algorithmic', float('nan')):.3f} |")
print()
for case, e in s.items():
    for k, v in (e.get("sq") or {}).items():
        sh = v["share_of_wave_cycles"]
        print(f"SQ {case}: {k[:60]}: wait_any {sh.get('SQ_WAIT_ANY')} wait_inst {sh.get('SQ_WAIT_INST_ANY')} active {sh.get('SQ_ACTIVE_INST_ANY')} valu {sh.get('SQ_ACTIVE_INST_VALU')}")
for case in ("epilogue", "epilogue_nb38", "band_zq", "band_cfg4"):
    e = s.get(case)
    if e:
        print(case, {k[:40]: round(v["avg_us"] / 1e3, 3) for k, v in e["kernels"].items()}, {k[:40]: v.get("hbm_bytes_per_launch") for k, v in (e.get("pmc") or {}).items()})
