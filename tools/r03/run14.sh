#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q > $O/gputest14.log 2>&1
tail -3 $O/gputest14.log
: > $O/ragged14.txt
for t in "" "--tune=2:128" "" "--tune=2:128"; do
  echo "== $t" >> $O/ragged14.txt
  timeout -k 10 300 python tools/ragged_sweep.py --schemes=4s,2s,bl $t 2>&1 | grep -v amdgpu.ids >> $O/ragged14.txt
done
cat $O/ragged14.txt
