#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not fullsize" 2>&1 | tail -5 | tee gpurun_out/r03/pytest_tripack.txt
{
for shape in "400000 12 60" "400000 16 60" "300000 24 60" "200000 32 60"; do
for t in "5:1" "5:0"; do
  echo "== $shape tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py $shape --schemes=zq,n79 --tune=$t 2>&1 | grep "n79"
done; done
} | tee gpurun_out/r03/tripack_n79.txt
