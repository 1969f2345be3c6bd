#!/bin/bash
# why do output sets of 17-94 GB run at 0.81-0.82 of the peak where 5.9-8.8 GB sets run at 0.88?  Address-translation
# (UTCL1 / UTCL2) and memory-side (TCC -> EA) counters of the 2s solve kernel at 1e4 columns (5.9 GB) and 125 000 columns (73.5 GB, the
# north-star per-GPU share), separate --pmc passes, program directly after `--`.  usage: bash tools/translation_counters.sh <outdir>
R=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-$R/gpurun_out/r03/translation}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PASSES=(
"utcl1|TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
"utcl1b|TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum"
"tcc|TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
"tcc2|TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_STREAMING_REQ_sum"
"grbm|GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
"tcp|TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
)
for ncol in 10000 125000; do
  for entry in "${PASSES[@]}"; do
    IFS='|' read -r name ctrs <<< "$entry"
    d=$O/raw_${ncol}_$name
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --scheme 2s --ncol $ncol --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-pcie --no-check > $O/${ncol}_$name.json 2> $O/${ncol}_$name.err || echo "pass $name failed at $ncol"
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$ncol" "$name" >> $O/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
dur = []
if sys.argv[1]:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'k_pipe' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(f"ncol {sys.argv[2]} pass {sys.argv[3]}: mean kernel {sum(dur)/max(len(dur),1)/1e3:.1f} us; " + "; ".join(f"{k} {sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
    rm -rf $d
  done
done
cat $O/summary.txt
