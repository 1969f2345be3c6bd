#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "13:2" "13:0" "13:2" "13:0"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79,zq --tune=$t 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r03/nb107_wl.txt
