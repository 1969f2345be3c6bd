#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "5:1" "5:0"; do
  echo "== tune $t"
  timeout -k 10 300 python3 tools/ragged_sweep.py 100000 38 100 --schemes=zq,n79 --tune=$t 2>&1 | grep "uniform\|ragged"
  timeout -k 10 300 python3 tools/ragged_sweep.py 150000 38 60 --schemes=zq,n79 --tune=$t 2>&1 | grep "uniform\|ragged"
  timeout -k 10 300 python3 tools/ragged_sweep.py 150000 36 60 --schemes=zq --tune=$t 2>&1 | grep "uniform"
  timeout -k 10 300 python3 tools/ragged_sweep.py 150000 42 60 --schemes=zq --tune=$t 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/tripack_38.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "not fullsize" 2>&1 | tail -3
