#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest17.log 2>&1
tail -3 $O/gputest17.log
( timeout -k 10 400 python tools/ragged_sweep.py --schemes=n79,zq,zq_pa,4s; timeout -k 10 200 python tools/ragged_sweep.py 6000 300 100 --schemes=zq,n79,zq_pa ) 2>&1 | grep -v amdgpu.ids > $O/ragged17.txt
cat $O/ragged17.txt
