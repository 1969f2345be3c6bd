#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "0:0" "2:2,4:2" "2:2,4:2,3:2" "2:2,4:2,3:4" "4:2" "2:2,4:3"; do
  echo "== tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=g77,bf --tune=$t 2>&1 | grep "uniform\|ragged"
done
} | tee gpurun_out/r03/g77_pipe_T2.txt
