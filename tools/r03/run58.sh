#!/bin/bash
mkdir -p gpurun_out/r03
{
for lib in "" skipl onlyl; do
  if [ -n "$lib" ]; then export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_$lib.so; else unset CRT1D_HIP_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 2>&1 | grep "uniform"
done
} | tee gpurun_out/r03/n79_layer_arrays_diag.txt
