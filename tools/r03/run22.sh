#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q -k "zq" > $O/gputest22.log 2>&1
tail -3 $O/gputest22.log
: > $O/zqpa22.txt
for t in "" "--tune=10:5"; do
  echo "== $t" >> $O/zqpa22.txt
  ( timeout -k 10 200 python tools/ragged_sweep.py --schemes=zq_pa,zq $t; timeout -k 10 200 python tools/ragged_sweep.py 6000 300 100 --schemes=zq_pa $t; timeout -k 10 200 python tools/ragged_sweep.py 100000 38 100 --schemes=zq_pa,zq $t;  timeout -k 10 200 python tools/ragged_sweep.py 30000 106 60 --schemes=zq_pa $t ) 2>&1 | grep -v amdgpu.ids | grep uniform >> $O/zqpa22.txt
done
cat $O/zqpa22.txt
for s in zq n79 2s; do timeout -k 10 100 python bench.py --scheme $s --variant integrated --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('integrated', d['config']['scheme'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel'])"; done
