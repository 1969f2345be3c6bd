#!/bin/bash
mkdir -p gpurun_out/r03
{
for t in "10:5" "10:0" "11:1" "11:3" "13:2"; do echo "== tune $t"; timeout -k 10 200 python3 tools/ragged_sweep.py 30000 107 60 --schemes=zq_pa --tune=$t 2>&1 | grep "uniform\|ragged"; done
timeout -k 10 200 python3 tools/ragged_sweep.py 20000 129 100 --schemes=zq_pa 2>&1 | grep "uniform"
timeout -k 10 200 python3 tools/ragged_sweep.py 20000 129 100 --schemes=zq_pa --tune=10:5 2>&1 | grep "uniform"
} | tee gpurun_out/r03/zqpa_odd_nb.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "zq_pa or odd or famil or bitwise" 2>&1 | tail -3
timeout -k 10 600 python3 tools/fuzz_parity.py 150 77 2>&1 | grep -v amdgpu.ids | grep -v ": ok" | tail -5
