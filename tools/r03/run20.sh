#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "zq_pa" > $O/gputest20.log 2>&1
tail -5 $O/gputest20.log
: > $O/zqpa20.txt
for t in "" "--tune=10:6" "--tune=10:7" "--tune=10:5"; do
  echo "== $t" >> $O/zqpa20.txt
  ( timeout -k 10 200 python tools/ragged_sweep.py --schemes=zq_pa $t; timeout -k 10 200 python tools/ragged_sweep.py 6000 300 100 --schemes=zq_pa $t; timeout -k 10 200 python tools/ragged_sweep.py 100000 38 100 --schemes=zq_pa $t ) 2>&1 | grep -v amdgpu.ids | grep uniform >> $O/zqpa20.txt
done
cat $O/zqpa20.txt
