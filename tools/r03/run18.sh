#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_band.py -m gpu -q -x > $O/gputest18.log 2>&1
tail -15 $O/gputest18.log
