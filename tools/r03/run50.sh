#!/bin/bash
mkdir -p gpurun_out/r03
export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stampz.so
timeout -k 10 200 python3 tools/stamp_timeline_tri.py zq_pa 10000 300 60 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/stamp_zqpa_now.txt
