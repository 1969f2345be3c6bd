#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "n79 or tridiag or famil or bitwise" 2>&1 | tail -8 | tee gpurun_out/r03/pytest_n79.txt
{
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=n79 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python3 tools/ragged_sweep.py 100000 38 60 --schemes=n79 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r03/n79_projective.txt
