#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
: > $O/ragged12.txt
for v in wpe5 wpe6; do
for t in "4:2,3:3" "4:2,3:2" "4:2,3:1" "4:4,3:3" "4:3,3:2"; do
  echo "== $v tune $t" >> $O/ragged12.txt
  CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_$v.so timeout -k 10 300 python tools/ragged_sweep.py --schemes=4s --tune=$t 2>&1 | grep -v amdgpu.ids >> $O/ragged12.txt
done
done
cat $O/ragged12.txt
