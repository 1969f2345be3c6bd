#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 500 python3 tools/ragged_sweep.py 10000 300 60 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python3 tools/ragged_sweep.py 6000 300 100 --schemes=n79,zq,zq_pa 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python3 tools/ragged_sweep.py 30000 107 60 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python3 tools/ragged_sweep.py 100000 38 100 --schemes=2s,4s,n79,zq,zq_pa 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 python3 tools/ragged_sweep.py 400000 12 60 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r03/ragged_sweep_final.txt | tail -5
