#!/bin/bash
for shape in "30000 106 60" "20000 128 60" "30000 100 60" "40000 66 60" "50000 64 60"; do timeout -k 10 200 python3 tools/ragged_sweep.py $shape --schemes=zq_pa 2>&1 | grep "uniform"; timeout -k 10 200 python3 tools/ragged_sweep.py $shape --schemes=zq_pa --tune=10:5 2>&1 | grep "uniform"; done
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "zq_pa" 2>&1 | tail -2
