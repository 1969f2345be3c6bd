#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=zq_pa,zq 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=zq_pa,zq 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=zq_pa,zq 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 100000 38 100 --schemes=zq_pa,zq 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 400000 12 60 --schemes=zq 2>&1 | grep "uniform"
python3 bench.py --scheme zq --variant integrated --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('zq integrated kernel_ms', r.get('kernel_ms_avg'), r.get('kernel'))"
} | tee gpurun_out/r03/zq_rcp_hoist.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not fullsize" 2>&1 | tail -2
