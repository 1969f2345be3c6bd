#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputest25.log 2>&1
tail -4 $O/gputest25.log
timeout -k 10 250 python bench.py > $O/bench_default_b.json 2> $O/bench_default_b.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_default_b.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d["check"]["ok"], d["check"]["max_rel_profile"], d["cpu_baseline"]["value"])
PY
