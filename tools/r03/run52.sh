#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 1100 python3 tools/fuzz_parity.py 500 303 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/fuzz_parity.txt | tail -15
