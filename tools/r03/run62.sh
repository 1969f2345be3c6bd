#!/bin/bash
for s in 4s zq_pa n79 zq; do
python3 bench.py --scheme $s --no-cpu-baseline --no-pcie --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$s', 'ms_per_step', round(d['ms_per_step'],4), 'kernel avg/med/min', round(r['kernel_ms_avg'],4), round(r['kernel_ms_median'],4), round(r['kernel_ms_min'],4), 'frac', round(r['frac'],3), 'k0', round(r['k0_ms'],4))
"
done
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=4s,zq_pa 2>&1 | grep uniform
