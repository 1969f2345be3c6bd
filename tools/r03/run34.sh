#!/bin/bash
O=$PWD/gpurun_out/r03/pmc_n79
rm -rf $O; mkdir -p $O
bash tools/pmc_groups.sh $O n79_nb107 k_tri_pipe icache,lds,mix,fetch,busy,grbm -- --scheme n79 --nb 107 --ncol 30000
bash tools/pmc_groups.sh $O n79_nb300 k_tri_pipe icache,lds,busy,grbm -- --scheme n79
cat $O/summary.txt
