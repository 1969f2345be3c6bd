#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/alloc_pmc.py > $R/gpurun_out/alloc_plain.txt 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_MULTI_MISS --kernel-trace --output-format csv -d $R/gpurun_out/alloc_pmc_tlb -- python3 $R/tools/alloc_pmc.py > $R/gpurun_out/alloc_pmc_tlb.txt 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY --kernel-trace --output-format csv -d $R/gpurun_out/alloc_pmc_tcc -- python3 $R/tools/alloc_pmc.py > $R/gpurun_out/alloc_pmc_tcc.txt 2>&1
