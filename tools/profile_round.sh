#!/bin/bash
# Round evidence (run on the GPU box from the repo root): rocprofv3 kernel-trace stats of the bench command for the headline
# (--placement none: every launch of the run then writes the same allocation, so that the rocprof average and the HIP-event average
#  of bench.py describe the same thing; with the placement search the stats would also average its slower candidate allocations)
# scheme and the two tridiagonal ones, then WRITE_SIZE / FETCH_SIZE in separate counter-only passes.  Output: gpurun_out/prof_*.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for S in 2s n79 zq; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$S -- python3 $R/bench.py --scheme $S --steps 20 --warmup 5 --no-cpu-baseline --placement none > $R/gpurun_out/prof_stats_$S.json 2> $R/gpurun_out/prof_stats_$S.err || exit 1
done
for S in 2s n79 zq; do
  for C in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc_${C}_$S -- python3 $R/bench.py --scheme $S --steps 3 --warmup 1 --no-cpu-baseline --placement none > $R/gpurun_out/prof_pmc_${C}_$S.log 2>&1 || exit 1
  done
done
