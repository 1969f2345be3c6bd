#!/bin/bash
mkdir -p gpurun_out/r03
O=$PWD/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -c "counter" $O/counters_list.txt
grep -i "utcl\|tlb\|translation\|ATC\|xnack" $O/counters_list.txt | head -60
