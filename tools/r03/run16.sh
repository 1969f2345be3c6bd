#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stamp.so
( timeout -k 10 120 python tools/stamp_timeline.py 4s 10000 300 60 --ragged ) 2>&1 | grep -v amdgpu.ids > $O/stamps16.txt
tail -22 $O/stamps16.txt
