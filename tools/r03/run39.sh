#!/bin/bash
mkdir -p gpurun_out/r03
export CRT1D_HIP_LIB=$PWD/variants/libcrt1d_hip_stampn.so
{
for r in "" "--ragged"; do for t in 13:2 13:0; do
echo "=== $r tune $t"
timeout -k 10 200 python3 tools/stamp_timeline_tri.py n79 30000 107 60 $r --tune=$t 2>&1 | grep -v amdgpu.ids
done; done
} | tee gpurun_out/r03/stamp_n79_wl.txt
