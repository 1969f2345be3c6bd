#!/bin/bash
mkdir -p gpurun_out/r03/final
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r03/final/pytest_gpu.txt
python3 bench.py > gpurun_out/r03/final/bench_default_n1.json 2> gpurun_out/r03/final/bench_default.err && tail -c 600 gpurun_out/r03/final/bench_default_n1.json
python3 bench.py --ragged --no-cpu-baseline > gpurun_out/r03/final/bench_ragged_n1.json 2>/dev/null
python3 bench.py --ncol 125000 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > gpurun_out/r03/final/bench_125k_columns_n1.json 2>/dev/null
python3 bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r03/final/bench_f32_n1.json 2>/dev/null
python3 bench.py --partition band --ncol 20000 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/r03/final/bench_band_n1_20000cols.json 2>/dev/null
python3 bench.py --scheme zq --nb 12 --ncol 400000 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > gpurun_out/r03/final/bench_zq_nb12.json 2>/dev/null
for f in gpurun_out/r03/final/bench_*.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], 'value %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'frac', r.get('frac'), 'kernel', r.get('kernel'), 'check', (d.get('check') or {}).get('ok'))
PY
done
