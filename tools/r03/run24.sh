#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
( timeout -k 10 300 python tools/ragged_sweep.py 30000 107 60; timeout -k 10 300 python tools/ragged_sweep.py 200000 38 60 --schemes=2s,4s,n79,zq,zq_pa;  timeout -k 10 300 python tools/ragged_sweep.py 400000 12 60 --schemes=2s,zq,n79; timeout -k 10 300 python tools/ragged_sweep.py 10000 300 60 --schemes=2s,n79,zq --dtype=f32 ) 2>&1 | grep -v amdgpu.ids | grep uniform > $O/shapes24.txt
cat $O/shapes24.txt
