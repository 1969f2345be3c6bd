#!/bin/bash
mkdir -p gpurun_out/r03
{
python3 bench.py --scheme 2s --dtype f32 --no-cpu-baseline --no-pcie --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench default steps', d['steps'], 'ms_per_step', d['ms_per_step'], {k:r[k] for k in r if k.startswith('kernel_ms') or k=='k0_ms'}, d['config']['output_placement']['classes'])
"
python3 tools/ragged_sweep.py 10000 300 60 --schemes=2s --dtype=f32 2>&1 | grep -v amdgpu
python3 bench.py --scheme 2s --dtype f32 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench 20 steps', d['ms_per_step'], {k:r[k] for k in r if k.startswith('kernel_ms') or k=='k0_ms'}, d['config']['output_placement']['classes'])
"
} 2>&1 | tee gpurun_out/r03/f32_bench_vs_sweep.txt
