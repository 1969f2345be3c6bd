#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q > $O/gputest10.log 2>&1
tail -3 $O/gputest10.log
timeout -k 10 400 python tools/ragged_sweep.py --schemes=2s,4s,bl,g77,bf 2>&1 | grep -v amdgpu.ids > $O/ragged10.txt
cat $O/ragged10.txt
