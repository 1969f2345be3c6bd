#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed_narrow" 2>&1 | tail -4 | tee gpurun_out/r03/pytest_mixed.txt
bash tools/profile_round.sh r03d '^(zq_nb12|2s_nb12|zq_nb8_wave|n79_nb12|4s_nb12|2s_f32|n79_f32)$' 2>&1 | tail -3
