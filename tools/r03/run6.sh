#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest6.log 2>&1
tail -3 $O/gputest6.log
timeout -k 10 400 python tools/ragged_sweep.py 2>&1 | grep -v amdgpu.ids > $O/ragged6.txt
cat $O/ragged6.txt
