#!/bin/bash
# Round evidence (run on the GPU box from the repo root): for every kernel the repo makes a claim about,
#   pass A  rocprofv3 --kernel-trace --stats          of the bench command (per-kernel average duration)
#   pass B  rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate pass, counters only)
#   pass C  rocprofv3 --pmc FETCH_SIZE --kernel-trace (separate pass; gfx950: doubled when summarised)
#   pass D  SQ busy / wait / VALU counters for the kernels that are NOT HBM-bound (zq_pa, f32 storage, integrated)
# The program follows `--` directly (python3 <script>); no traces are combined with --pmc.  Output: gpurun_out/prof_<tag>/<case>/<pass>/.
# usage: bash tools/profile_round.sh <tag> [case-name-filter]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
ONLY=${2:-}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie"
# name | pmc passes wanted (wf = WRITE+FETCH, sq = + SQ pass) | command after `python3`
CASES=(
"2s|wf|$R/bench.py --scheme 2s $B"
"4s|wf|$R/bench.py --scheme 4s $B"
"bl|wf|$R/bench.py --scheme bl $B"
"g77|wf|$R/bench.py --scheme g77 $B"
"bf|wf|$R/bench.py --scheme bf $B"
"n79|wf|$R/bench.py --scheme n79 $B"
"zq|wf|$R/bench.py --scheme zq $B"
"zq_pa|wfsq|$R/bench.py --scheme zq_pa $B"
"zq_nz100|wfsq|$R/bench.py --scheme zq --nz 100 --ncol 6000 $B"
"2s_f32|wfsq|$R/bench.py --scheme 2s --dtype f32 $B"
"n79_f32|wfsq|$R/bench.py --scheme n79 --dtype f32 $B"
"2s_nb107|wf|$R/bench.py --scheme 2s --nb 107 --ncol 30000 $B"
"n79_nb107|wfsq|$R/bench.py --scheme n79 --nb 107 --ncol 30000 $B"
"zq_nb107|wfsq|$R/bench.py --scheme zq --nb 107 --ncol 30000 $B"
"zq_nb38_nz100|wf|$R/bench.py --scheme zq --nb 38 --nz 100 --ncol 100000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"2s_nb38|wf|$R/bench.py --scheme 2s --nb 38 --ncol 200000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"zq_nb12|wf|$R/bench.py --scheme zq --nb 12 --ncol 400000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"2s_nb12|wf|$R/bench.py --scheme 2s --nb 12 --ncol 400000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"zq_nb8_wave|wfsq|$R/bench.py --scheme zq --nb 8 --ncol 400000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"n79_nb12|wf|$R/bench.py --scheme n79 --nb 12 --ncol 400000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"4s_nb12|wf|$R/bench.py --scheme 4s --nb 12 --ncol 400000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"2s_integrated|wfsq|$R/bench.py --scheme 2s --variant integrated $B"
"n79_integrated|wfsq|$R/bench.py --scheme n79 --variant integrated $B"
"zq_integrated|wfsq|$R/bench.py --scheme zq --variant integrated $B"
"band_zq|wf|$R/bench.py --partition band --ncol 20000 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline"
"2s_125k|wf|$R/bench.py --scheme 2s --ncol 125000 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-pcie"
"band_cfg4|wf|$R/bench.py --partition band --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline"
"epilogue|wf|$R/tools/epilogue_bench.py 10000 300 60"
"epilogue_nb38|wfsqi|$R/tools/epilogue_bench.py 100000 38 100"
"zq_pa_nb38|wfsq|$R/bench.py --scheme zq_pa --nb 38 --nz 100 --ncol 100000 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-pcie"
"zq_pa_nz100|wf|$R/bench.py --scheme zq_pa --nz 100 --ncol 6000 $B"
"n79_nz100|wf|$R/bench.py --scheme n79 --nz 100 --ncol 6000 $B"
"zq_pa_nb107|wfsq|$R/bench.py --scheme zq_pa --nb 107 --ncol 30000 $B"
# non-uniform dLAI (any strictly decreasing lai is legal input, crt1d/model.py:240-246): none of the uniform-column fast paths applies
"2s_ragged|wfsq|$R/bench.py --scheme 2s --ragged $B"
"4s_ragged|wfsqi|$R/bench.py --scheme 4s --ragged $B"
"bl_ragged|wf|$R/bench.py --scheme bl --ragged $B"
"g77_ragged|wfsq|$R/bench.py --scheme g77 --ragged $B"
"bf_ragged|wfsq|$R/bench.py --scheme bf --ragged $B"
"n79_ragged|wfsq|$R/bench.py --scheme n79 --ragged $B"
"zq_ragged|wf|$R/bench.py --scheme zq --ragged $B"
"zq_pa_ragged|wf|$R/bench.py --scheme zq_pa --ragged $B"
"zq_nz100_ragged|wf|$R/bench.py --scheme zq --nz 100 --ncol 6000 --ragged $B"
"n79_nz100_ragged|wfsq|$R/bench.py --scheme n79 --nz 100 --ncol 6000 --ragged $B"
)
for entry in "${CASES[@]}"; do
  IFS='|' read -r name passes cmd <<< "$entry"
  if [ -n "$ONLY" ] && ! [[ "$name" =~ $ONLY ]]; then continue; fi
  d=$OUT/$name
  mkdir -p $d
  echo "== $name" | tee -a $OUT/progress.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 $cmd > $d/stats.json 2> $d/stats.err || { echo "stats pass failed for $name" | tee -a $OUT/progress.log; continue; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $d/pmc_WRITE_SIZE -- python3 $cmd > $d/pmc_WRITE_SIZE.log 2>&1 || echo "WRITE_SIZE pass failed for $name" | tee -a $OUT/progress.log
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d/pmc_FETCH_SIZE -- python3 $cmd > $d/pmc_FETCH_SIZE.log 2>&1 || echo "FETCH_SIZE pass failed for $name" | tee -a $OUT/progress.log
  if [[ "$passes" == *sq* ]]; then
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $d/pmc_SQ -- python3 $cmd > $d/pmc_SQ.log 2>&1 || echo "SQ pass failed for $name" | tee -a $OUT/progress.log
  fi
  if [[ "$passes" == *sqi* ]]; then  # instruction counts
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $d/pmc_SQI -- python3 $cmd > $d/pmc_SQI.log 2>&1 || echo "SQI pass failed for $name" | tee -a $OUT/progress.log
  fi
done
python3 $R/tools/summarize_profiles.py $OUT > $OUT/summary.json 2> $OUT/summary.err
# the raw rocprofv3 output is too large to travel back: keep the summary + the trimmed evidence, drop the rest
SMALL=$R/gpurun_out/prof_${TAG}_small
rm -rf $SMALL && mkdir -p $SMALL
python3 $R/tools/summarize_profiles.py $OUT --shrink $SMALL
cp $OUT/summary.json $OUT/summary.err $OUT/progress.log $SMALL/
rm -rf $OUT
echo "done"
