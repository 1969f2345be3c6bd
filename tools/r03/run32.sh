#!/bin/bash
mkdir -p gpurun_out/r03
CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stampn.so timeout -k 10 200 python3 tools/stamp_timeline_tri.py n79 30000 107 60 2>&1 | tee gpurun_out/r03/stamp_n79_nb107.txt
CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stampn.so timeout -k 10 200 python3 tools/stamp_timeline_tri.py n79 10000 300 60 2>&1 | tee gpurun_out/r03/stamp_n79_nb300.txt
