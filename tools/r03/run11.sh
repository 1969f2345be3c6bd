#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/ragged11.txt
for t in "4:2,3:3" "4:4,3:2" "4:4,3:3" "4:4,3:4" "4:6,3:3" "4:8,3:3" "4:8,3:4" "4:3,3:3" "4:4,3:5"; do
  echo "== tune $t" >> $O/ragged11.txt
  timeout -k 10 300 python tools/ragged_sweep.py --schemes=4s --tune=$t 2>&1 | grep -v amdgpu.ids >> $O/ragged11.txt
done
cat $O/ragged11.txt
