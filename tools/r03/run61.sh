#!/bin/bash
mkdir -p gpurun_out/r03/final
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-device --ncol 4000 --steps 5 --warmup 2 --repeats 2 --no-cpu-baseline > gpurun_out/r03/final/bench_n2_gloo_shared_gpu.json 2> gpurun_out/r03/final/bench_n2.err; echo "rc=$?"; tail -c 700 gpurun_out/r03/final/bench_n2_gloo_shared_gpu.json; echo
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --share-device --partition band --ncol 4000 --steps 3 --warmup 1 --repeats 2 --no-cpu-baseline > gpurun_out/r03/final/bench_band_n2_gloo_shared_gpu.json 2> gpurun_out/r03/final/bench_band_n2.err; echo "rc=$?"; tail -c 900 gpurun_out/r03/final/bench_band_n2_gloo_shared_gpu.json; echo
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-pg --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline > gpurun_out/r03/final/bench_rccl_world1_column.json 2> gpurun_out/r03/final/bench_rccl1.err; echo "rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/r03/final/bench_rccl_world1_column.json').read().strip().splitlines()[-1]);print(d['value'], d.get('rccl'))"
