#!/bin/bash
# round-3 GPU step 1: full-size tests + buffer tests + bench lines with the self-check
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -x -q > $O/gputest1.log 2>&1
tail -5 $O/gputest1.log
timeout -k 10 250 python bench.py > $O/bench_default_a.json 2> $O/bench_default_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_default_a.json"))
print(d["value"], d["ms_per_step"], d["check"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
timeout -k 10 250 python bench.py --partition band --ncol 20000 --no-cpu-baseline > $O/bench_band_a.json 2> $O/bench_band_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_band_a.json"))
print(d["value"], d["check"], d["check_integrated"])
PY
