#!/bin/bash
mkdir -p gpurun_out/r03
{
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 --tune=13:1 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=n79 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=n79 --tune=13:1 2>&1 | grep "uniform\|ragged"
timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 --dtype=f32 2>&1 | grep "uniform\|ragged"
} | tee gpurun_out/r03/n79_rotated_layer_stores.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "n79 or famil or bitwise or tridiag" 2>&1 | tail -2
