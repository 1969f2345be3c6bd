#!/bin/bash
mkdir -p gpurun_out/r03/noprof
timeout -k 10 1100 python3 tools/bench_cases_noprof.py gpurun_out/r03/noprof 2>&1 | grep -v amdgpu.ids | tail -50
