#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "13:2" "13:3" "13:3,11:3" "13:3,11:4" "13:2,11:3" "13:3,11:1"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79 --tune=$t 2>&1 | grep "uniform\|ragged"
done
} | tee gpurun_out/r03/n79_wl_storewaves.txt
