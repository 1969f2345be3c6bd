#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 300 python3 tools/f32_levels_check.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=2s,bl --dtype=f32 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=2s,bl --dtype=f32a 2>&1 | grep -v amdgpu.ids
for t in 4:8 4:4 "4:4,3:2" "4:8,3:2" "4:2"; do echo "== tune $t"; timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=2s --dtype=f32a --tune=$t 2>&1 | grep uniform; done
} | tee gpurun_out/r03/f32_levels.txt
