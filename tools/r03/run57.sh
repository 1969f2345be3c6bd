#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "10:0" "10:7" "10:6"; do
echo "== tune $t"
timeout -k 10 200 python3 tools/ragged_sweep.py 100000 38 100 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 30000 106 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 20000 128 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 50000 64 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/zqpa_pipe2_widths.txt
