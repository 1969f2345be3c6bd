#!/bin/bash
# Counter groups for one bench command, one rocprofv3 --pmc pass per group (program directly after `--`), per-launch means of the kernels
# whose name contains <match>.   usage: bash tools/pmc_groups.sh <outdir> <tag> <match> <groups: icache,lds,mix,fetch,...> -- <bench.py args>
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; TAG=$2; MATCH=$3; GROUPS_=$4; shift 5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A G
G[icache]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
G[lds]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
G[mix]="SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_VALU_INT32"
G[fetch]="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
G[busy]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVES"
G[grbm]="GRBM_GUI_ACTIVE"
for name in ${GROUPS_//,/ }; do
  d=$O/raw_${TAG}_$name
  rocprofv3 --pmc ${G[$name]} --kernel-trace --output-format csv -d $d -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-pcie --no-check > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || echo "pass $name failed"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$TAG" "$name" "$MATCH" >> $O/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
dur = []
if sys.argv[1]:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[4] in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
n = max(len(acc), 1)
print(f"{sys.argv[2]} [{sys.argv[3]}] kernel {sum(dur)/max(len(dur),1)/1e3:.1f} us: " + "; ".join(f"{k} {sum(v)/len(v):.5g}" for k, v in sorted(acc.items())))
PY
  rm -rf $d
done
