#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not fullsize" 2>&1 | tail -5 | tee gpurun_out/r03/pytest_pack.txt
{
for t in "5:1" "5:0" "4:8" "6:2" "6:2,3:2" "6:2,4:8" "6:3,3:1" "6:3,3:1,4:8"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 400000 12 60 --schemes=2s,bl,4s,g77 --tune=$t 2>&1 | grep -v amdgpu.ids | grep "uniform\|ragged"
done
} | tee gpurun_out/r03/pack_nb12.txt
