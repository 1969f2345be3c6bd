#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 300 python3 tools/sustained_probe.py 2s 2>&1 | tee gpurun_out/r03/sustained_2s.txt
