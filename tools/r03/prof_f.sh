#!/bin/bash
bash tools/profile_round.sh r03f '^(zq_pa_nb107)$' 2>&1 | tail -2
mkdir -p gpurun_out/r03/noprof2
timeout -k 10 300 python3 tools/bench_cases_noprof.py gpurun_out/r03/noprof2 '^zq_pa_nb107$' 2>&1 | grep -v amdgpu.ids | tail -2
