#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/ragged7.txt
for t in "" "--tune=2:64" "" "--tune=2:64"; do
  echo "== $t" >> $O/ragged7.txt
  timeout -k 10 300 python tools/ragged_sweep.py --schemes=2s,4s $t 2>&1 | grep -v amdgpu.ids >> $O/ragged7.txt
done
cat $O/ragged7.txt
