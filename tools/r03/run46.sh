#!/bin/bash
mkdir -p gpurun_out/r03
{
for shape in "600000 4 60" "600000 8 60" "400000 16 60" "300000 24 60" "200000 32 60"; do
for t in "5:1" "5:0"; do
  echo "== $shape tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py $shape --schemes=2s,4s,g77 --tune=$t 2>&1 | grep "uniform"
done; done
for t in "5:1" "5:2" "5:2,6:3" "5:2,3:2" "5:2,6:3,3:2" "5:2,6:4,3:2"; do
  echo "== 200000 38 60 tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 200000 38 60 --schemes=2s,4s,g77 --tune=$t 2>&1 | grep "uniform"
done
for t in "5:1" "5:2,6:3" "5:2,6:4,3:2"; do
  echo "== 100000 64 60 tune $t"
  timeout -k 10 200 python3 tools/ragged_sweep.py 100000 64 60 --schemes=2s,4s --tune=$t 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/pack_widths.txt
