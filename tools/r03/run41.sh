#!/bin/bash
mkdir -p gpurun_out/r03
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03/pytest_full.txt
