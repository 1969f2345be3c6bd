#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "9:4" "9:2" "9:2,11:1" "9:4,11:1" "9:2,11:3"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79,zq --tune=$t 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r03/nb107_T2.txt
