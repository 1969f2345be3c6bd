#!/bin/bash
O=$PWD/gpurun_out/r03/pmc_zqpa
rm -rf $O; mkdir -p $O
bash tools/pmc_groups.sh $O zq_pa k_zqpa icache,lds,mix,busy,fetch,grbm -- --scheme zq_pa
bash tools/pmc_groups.sh $O zq k_tri_pipe lds,mix,busy,grbm -- --scheme zq
cat $O/summary.txt
