#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "0:0" "8:8" "11:2" "8:8,11:2" "10:6" "11:4"; do
echo "== tune $t"
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/zqpa_tune.txt
