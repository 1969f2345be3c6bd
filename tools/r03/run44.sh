#!/bin/bash
mkdir -p gpurun_out/r03
{
for lib in "" split ""  split; do
  if [ -n "$lib" ]; then export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_$lib.so; else unset CRT1D_HIP_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/ragged_sweep.py 10000 300 60 --schemes=n79 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python3 tools/ragged_sweep.py 6000 300 100 --schemes=n79 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r03/n79_split_stores.txt
