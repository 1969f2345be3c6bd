"""Run every bench.py case of tools/profile_round.sh WITHOUT the profiler and keep the bench lines: the HIP-event kernel time of the same
commands whose rocprofv3 averages fill DESIGN.md section 5 (rocprofv3 runs the compute-heavy kernels 10 % slower; the HBM-bound ones 1-2 %).

    python tools/bench_cases_noprof.py <outdir> [case-regex]"""
import json
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
only = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
os.makedirs(out, exist_ok=True)
src = open(os.path.join(root, "tools", "profile_round.sh")).read()
B = re.search(r'^B="([^"]*)"', src, re.M).group(1)
res = {}
for name, _, cmd in re.findall(r'^"([a-z0-9_]+)\|([a-z]+)\|([^"]*)"', src, re.M):
    if "bench.py" not in cmd or (only and not only.search(name)):
        continue
    args = cmd.replace("$R/bench.py", "").replace("$B", B).split()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--no-check"], capture_output=True, text=True, cwd=root)
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    try:
        d = json.loads(line)
    except Exception:
        print(name, "FAILED", p.stderr[-300:], flush=True)
        continue
    open(os.path.join(out, name + ".json"), "w").write(line + "\n")
    r = d.get("roofline") or {}
    res[name] = {"kernel": r.get("kernel"), "kernel_ms_avg": r.get("kernel_ms_avg"), "kernel_ms_min": r.get("kernel_ms_min"), "frac": r.get("frac"),
                 "ms_per_step": d.get("ms_per_step"), "value": d.get("value")}
    print(name, res[name], flush=True)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
