#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/ragged_ab.txt
for rep in 1 2; do
for v in "" ab1 ab2; do
  echo "== variant '${v}' rep $rep" >> $O/ragged_ab.txt
  if [ -n "$v" ]; then export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_$v.so; else unset CRT1D_HIP_LIB; fi
  timeout -k 10 200 python tools/ragged_sweep.py --schemes=4s,g77,bf 2>&1 | grep -v amdgpu.ids >> $O/ragged_ab.txt
done
done
cat $O/ragged_ab.txt
