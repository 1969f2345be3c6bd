#!/bin/bash
# Evidence for the output placement: N consecutive PROCESSES of the default bench workload with the set
# allocator (--placement auto) and with torch.empty (--placement none); one JSON line each -> gpurun_out/placement_processes_<tag>.jsonl
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-10}
TAG=${2:-r02}
S=${3:-2s}
OUT=$R/gpurun_out/placement_processes_${TAG}_$S.jsonl
: > $OUT
for i in $(seq 1 $N); do
  for P in auto none; do
    timeout -k 10 120 python3 $R/bench.py --scheme $S --placement $P --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-pcie 2>/dev/null | grep '^{' >> $OUT || exit 1
  done
done
python3 - $OUT <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for p in ("auto", "none"):
    sel = [r for r in rows if (r["config"]["output_placement"] != "torch.empty (--placement none)") == (p == "auto")]
    k = sorted(r["roofline"]["kernel_ms_avg"] for r in sel)
    s = sorted(r["ms_per_step"] for r in sel)
    print(f"{p:5s} n={len(sel)} kernel_ms min/median/max {k[0]:.4f} {k[len(k)//2]:.4f} {k[-1]:.4f} | ms_per_step min/median/max {s[0]:.4f} {s[len(s)//2]:.4f} {s[-1]:.4f}")
PY
