#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/ragged8.txt
for shape in "11719 256 60" "9375 320 60" "15625 192 60" "10000 300 60"; do
  timeout -k 10 300 python tools/ragged_sweep.py $shape --schemes=4s 2>&1 | grep -v amdgpu.ids >> $O/ragged8.txt
done
cat $O/ragged8.txt
