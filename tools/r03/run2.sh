#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/gputest2.log 2>&1
tail -5 $O/gputest2.log
timeout -k 10 300 python tools/ragged_sweep.py --schemes=2s,4s,g77,bf > $O/ragged1.txt 2>&1
cat $O/ragged1.txt
