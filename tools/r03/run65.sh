#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "10:0" "10:7,11:2" "10:7,11:1" "10:6,11:2" "10:6,11:1" "10:7,11:2,8:8" "10:6,11:1,8:8"; do
echo "== tune $t"
timeout -k 10 200 python3 tools/ragged_sweep.py 30000 106 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 100000 38 100 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/zqpa_pipe2_narrow_tune.txt
