#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q > $O/gputest4.log 2>&1
tail -5 $O/gputest4.log
: > $O/ragged4.txt
for t in "" "--tune=2:32"; do
  echo "== $t" >> $O/ragged4.txt
  timeout -k 10 300 python tools/ragged_sweep.py --schemes=2s,4s,g77,bf,bl $t 2>&1 | grep -v amdgpu.ids >> $O/ragged4.txt
done
cat $O/ragged4.txt
