#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stamp.so
( timeout -k 10 120 python tools/stamp_timeline.py 4s 10000 300 60 --ragged; timeout -k 10 120 python tools/stamp_timeline.py 4s 10000 300 60; timeout -k 10 120 python tools/stamp_timeline.py 2s 10000 300 60 ) 2>&1 | grep -v amdgpu.ids > $O/stamps15.txt
cat $O/stamps15.txt
