#!/bin/bash
mkdir -p gpurun_out/r03
for s in 2s n79 zq; do
python3 bench.py --scheme $s --nb 107 --ncol 30000 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$s', r['kernel'], 'frac', round(r['frac'],3), 'kernel_ms', round(r['kernel_ms_avg'],3), 'store_set', r['measured_store_set_GBs'], 'frac_of_store_set', r['frac_of_measured_store_set'])
"
done 2>&1 | tee gpurun_out/r03/nb107_store_set.txt
