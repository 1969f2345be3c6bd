#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 400 python3 tools/subrange_probe.py 2s 2>&1 | tee gpurun_out/r03/subrange_2s.txt
