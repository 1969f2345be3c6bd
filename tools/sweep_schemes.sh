#!/bin/bash
# bench every scheme at the config-2 shape (and zq at nz=100, f32, zq_pa); one JSON line each
for s in 2s 4s bl g77 bf n79 zq zq_pa; do
  timeout -k 10 200 python bench.py --scheme $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null
done
timeout -k 10 200 python bench.py --scheme zq --nz 100 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null
timeout -k 10 200 python bench.py --scheme 2s --nb 107 --ncol 30000 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null
timeout -k 10 200 python bench.py --scheme n79 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null
timeout -k 10 200 python bench.py --scheme 2s --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null
