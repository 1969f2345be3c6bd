#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_round.sh: per case the dominant crt:: kernel's average duration (kernel-trace stats),
HBM bytes per launch from the separate --pmc passes (WRITE_SIZE exact, FETCH_SIZE doubled: gfx950 counts 128-B requests at 64 B,
/opt/skills/guides/MI355X_MICROARCH.md section HBM; counter values are KiB), and SQ busy/wait shares where collected.
usage: summarize_profiles.py <prof dir>   -> JSON on stdout (also small CSV copies next to it for profiles/)"""
import csv
import glob
import json
import os
import sys

SKIP = ("k_fill", "k_copy", "k_probe_cols", "k_colpre", "__amd_rocclr", "at::native", "Cijk", "elementwise", "vectorized")


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.replace("crt::(anonymous namespace)::", "").replace("crt::", "").replace("void ", "")
    return n.split("(")[0] if len(n) > 150 else n


def main(root):
    out = {}
    for case in sorted(os.listdir(root)):
        d = os.path.join(root, case)
        if not os.path.isdir(d):
            continue
        e = {}
        st = find(os.path.join(d, "stats"), "*kernel_stats.csv")
        kernels = {}
        if st:
            for r in csv.DictReader(open(st)):
                kernels[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                      "max_us": float(r["MaxNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
        ours = {k: v for k, v in kernels.items() if not any(s in k for s in SKIP)}
        e["kernels"] = {short(k): v for k, v in sorted(ours.items(), key=lambda kv: -kv[1]["total_ms"])}
        dom = max(ours, key=lambda k: ours[k]["total_ms"]) if ours else None
        e["dominant"] = short(dom) if dom else None
        try:
            line = [ln for ln in open(os.path.join(d, "stats.json")) if ln.startswith("{")]
            if line:
                b = json.loads(line[0])
                if "roofline" in b:
                    e["bench"] = {k: b["roofline"].get(k) for k in ("kernel", "kernel_ms_avg", "achieved", "frac", "algorithmic_bytes_per_launch", "launch_shape", "frac_of_measured_store_set")}
                    e["bench"]["ms_per_step"] = b["ms_per_step"]
                else:
                    e["bench"] = b
        except Exception:
            pass
        for cname in ("WRITE_SIZE", "FETCH_SIZE"):
            f = find(os.path.join(d, "pmc_" + cname), "*counter_collection.csv")
            if not f:
                continue
            per = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname:
                    continue
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            for k, v in per.items():
                if any(s in k for s in SKIP):
                    continue
                kk = short(k)
                e.setdefault("pmc", {}).setdefault(kk, {})[cname + "_KiB_mean"] = sum(v) / len(v)
        for kk, c in e.get("pmc", {}).items():
            w, f = c.get("WRITE_SIZE_KiB_mean"), c.get("FETCH_SIZE_KiB_mean")
            if w is not None and f is not None:
                c["hbm_bytes_per_launch"] = int(w * 1024 + 2 * f * 1024)
        f = find(os.path.join(d, "pmc_SQ"), "*counter_collection.csv")
        if f:
            acc = {}
            for r in csv.DictReader(open(f)):
                if any(s in r["Kernel_Name"] for s in SKIP):
                    continue
                acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            sq = {}
            for k, c in acc.items():
                m = {n: sum(v) / len(v) for n, v in c.items()}
                wc = m.get("SQ_WAVE_CYCLES") or 1.0
                sq[k] = {"means": m, "share_of_wave_cycles": {n: round(m[n] / wc, 3) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS") if n in m}}
            e["sq"] = sq
        f = find(os.path.join(d, "pmc_SQI"), "*counter_collection.csv")
        if f:  # instruction counts per launch
            acc = {}
            for r in csv.DictReader(open(f)):
                if any(s in r["Kernel_Name"] for s in SKIP):
                    continue
                acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            e["sq_insts"] = {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in acc.items()}
        if dom and "bench" in e and isinstance(e["bench"], dict) and e["bench"].get("algorithmic_bytes_per_launch"):
            alg = e["bench"]["algorithmic_bytes_per_launch"]
            e["rocprof_avg_ms"] = ours[dom]["avg_us"] / 1e3
            e["rocprof_GBs"] = alg / (ours[dom]["avg_us"] * 1e-6) / 1e9
            e["rocprof_frac_of_8TBs"] = e["rocprof_GBs"] / 8000.0
            p = e.get("pmc", {}).get(short(dom), {})
            if "hbm_bytes_per_launch" in p:
                e["traffic_over_algorithmic"] = p["hbm_bytes_per_launch"] / alg
        out[case] = e
    json.dump(out, sys.stdout, indent=1)
    return out


def traffic_entries(summary, source):
    """profiles/pmc_traffic.json entries (read by bench.py into roofline.traffic) from a summary."""
    ents = []
    for case, e in summary.items():
        b = e.get("bench") or {}
        dom = e.get("dominant")
        p = (e.get("pmc") or {}).get(dom) or {}
        if not b.get("launch_shape") or "hbm_bytes_per_launch" not in p:
            continue
        ents.append({"case": case, "kernel_reported_by_library": b.get("kernel"), "kernel_rocprof": dom, "shape": b["launch_shape"],
                     "WRITE_SIZE_KiB": p.get("WRITE_SIZE_KiB_mean"), "FETCH_SIZE_KiB_raw": p.get("FETCH_SIZE_KiB_mean"),
                     "hbm_bytes_per_launch": p["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": b.get("algorithmic_bytes_per_launch"),
                     "ratio": round(p["hbm_bytes_per_launch"] / b["algorithmic_bytes_per_launch"], 4) if b.get("algorithmic_bytes_per_launch") else None,
                     "rocprof_avg_ms": e.get("rocprof_avg_ms"), "source": f"{source}/{case}/pmc_WRITE_SIZE.csv, pmc_FETCH_SIZE.csv"})
    return ents


def shrink(root, dst):
    """Copy the evidence worth committing (kernel stats, the bench line of the profiled run, the counter rows of this library's
    kernels) from a raw profile_round.sh directory into `dst` -- the raw rocprofv3 output exceeds what gpurun merges back."""
    import shutil

    for case in sorted(os.listdir(root)):
        src = os.path.join(root, case)
        if not os.path.isdir(src):
            continue
        out = os.path.join(dst, case)
        os.makedirs(out, exist_ok=True)
        st = find(os.path.join(src, "stats"), "*kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(out, "kernel_stats.csv"))
        sj = os.path.join(src, "stats.json")
        if os.path.exists(sj):
            lines = [ln for ln in open(sj) if ln.startswith("{")]
            if lines:
                open(os.path.join(out, "bench_line.json"), "w").write(lines[0])
        for c in ("WRITE_SIZE", "FETCH_SIZE", "SQ", "SQI"):
            f = find(os.path.join(src, "pmc_" + c), "*counter_collection.csv")
            if f:
                rows = open(f).read().splitlines()
                keep = [rows[0]] + [r for r in rows[1:] if "crt::" in r and "k_fill" not in r and "k_copy" not in r and "k_probe" not in r][:400]
                open(os.path.join(out, f"pmc_{c}.csv"), "w").write("\n".join(keep) + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--shrink":
        shrink(sys.argv[1], sys.argv[3])
    elif len(sys.argv) > 3 and sys.argv[2] == "--emit-traffic":
        # usage: summarize_profiles.py <merged summary.json> --emit-traffic <out.json> <source label>
        summ = json.load(open(sys.argv[1]))
        doc = {"_comment": "HBM bytes per launch of each solve kernel from rocprofv3 --pmc (separate passes WRITE_SIZE / FETCH_SIZE of the bench "
                           "command, tools/profile_round.sh). Counter values are KiB. gfx950 correction per MI355X_MICROARCH.md section HBM: "
                           "FETCH_SIZE reports 1/2 of a wide coalesced read -> doubled. bench.py looks an entry up by (kernel family as the library reports it, launch shape).",
               "entries": traffic_entries(summ, sys.argv[4] if len(sys.argv) > 4 else "profiles")}
        json.dump(doc, open(sys.argv[3], "w"), indent=1)
    else:
        main(sys.argv[1])
