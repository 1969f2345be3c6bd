#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "n79 or zq or tridiag or famil or bitwise or odd or generic" 2>&1 | tail -8 | tee gpurun_out/r03/pytest_wl.txt
{
for t in "13:2" "13:0"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=n79,zq --tune=$t 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r03/nb107_wl.txt
