#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "13:0" "14:1024,15:256" "14:1024,15:1" "14:1024,15:8" "14:1024,15:32" "14:2048,15:256" "14:2048,15:1" "14:2048,15:8"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79,zq --tune=$t 2>&1 | grep -v amdgpu.ids | grep uniform
done
} | tee gpurun_out/r03/nb107_stagger.txt
