#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 400 python3 tools/ragged_sweep.py 10000 300 60 --dtype=f32 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python3 tools/ragged_sweep.py 6000 300 100 --dtype=f32 --schemes=n79,zq,zq_pa 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r03/f32_all.txt
