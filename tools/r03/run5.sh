#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q > $O/gputest5.log 2>&1
tail -3 $O/gputest5.log
timeout -k 10 300 python tools/ragged_sweep.py --schemes=2s,4s,g77,bf,bl 2>&1 | grep -v amdgpu.ids > $O/ragged5.txt
cat $O/ragged5.txt
bash tools/profile_round.sh r03a '^4s_ragged$|^4s$' > $O/prof5.log 2>&1
tail -3 $O/prof5.log
