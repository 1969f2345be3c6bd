#!/bin/bash
mkdir -p gpurun_out/r03/clk
O=$PWD/gpurun_out/r03/clk
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="--steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie --no-check"
for c in "4s --ragged" "4s" "2s"; do
  n=$(echo $c | tr -d ' -')
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/raw_$n -- python3 $R/bench.py --scheme $c $B > $O/$n.json 2> $O/$n.err
  f=$(find $O/raw_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pipe' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        acc['v'].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp']))))
v = acc['v'][5:]
clk = [a / 8 / (t * 1e-9) / 1e9 for a, t in v]
print(sys.argv[2], "n", len(v), "mean dur us", sum(t for _, t in v) / len(v) / 1e3, "eff clock GHz mean", sum(clk) / len(clk), "min", min(clk), "max", max(clk))
PY
  rm -rf $O/raw_$n
done
