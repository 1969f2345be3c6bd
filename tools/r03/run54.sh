#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=zq_pa 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=zq_pa 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=zq_pa --tune=10:5 2>&1 | grep "uniform\|ragged"
} | tee gpurun_out/r03/zqpa_prefetch.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "zq_pa" 2>&1 | tail -2
