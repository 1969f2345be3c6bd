#!/bin/bash
mkdir -p gpurun_out/r03/sizes
for n in 10000 15000 20000 30000 50000 80000 125000; do
  python3 bench.py --scheme 2s --ncol $n --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-pcie --no-check > gpurun_out/r03/sizes/$n.json 2> gpurun_out/r03/sizes/$n.err || echo fail $n
  python3 - gpurun_out/r03/sizes/$n.json <<'PY'
import json, sys, collections
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
cl = d['config']['output_placement']['classes']
print(d['config']['ncol_per_gpu'], 'kernel_ms', round(r['kernel_ms_avg'], 3), 'frac', round(r['frac'], 4), 'fill_set', round(r['measured_store_set_GBs']), 'lin', round(r['probe_linear_fill_GBs']),
      {k: dict(collections.Counter(v)) for k, v in cl.items()})
for k, v in cl.items(): print('   ', k, v)
PY
done
