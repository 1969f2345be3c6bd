#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/zqpa23.txt
for t in "--tune=8:8,11:3" "--tune=8:8,11:2,10:6" ""; do
  echo "== $t" >> $O/zqpa23.txt
  ( timeout -k 10 200 python tools/ragged_sweep.py --schemes=zq_pa $t; timeout -k 10 200 python tools/ragged_sweep.py 6000 300 100 --schemes=zq_pa $t ) 2>&1 | grep -v amdgpu.ids | grep uniform >> $O/zqpa23.txt
done
cat $O/zqpa23.txt
