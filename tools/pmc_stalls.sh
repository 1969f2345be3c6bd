#!/bin/bash
# Stall-reason counters for one scheme's solve kernel (separate passes; counters only, no API traces).
# usage (on the GPU box, from the repo root): bash tools/pmc_stalls.sh n79
S=${1:-n79}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${S}_${name} -- python3 $R/bench.py --scheme $S --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${S}_${name}.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES &&
run sq2 SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR SQ_WAVES &&
run tcc TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_BUSY &&
run tcp TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_TCP_TA_DATA_STALL_CYCLES TA_BUSY
