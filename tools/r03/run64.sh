#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 500 python3 tools/robust_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/robust_sweep.log | tail -12
timeout -k 10 600 python3 tools/fuzz_parity.py 400 911 2>&1 | grep -v amdgpu.ids | grep -v ": ok" | tee gpurun_out/r03/fuzz_parity_2.txt | tail -8
