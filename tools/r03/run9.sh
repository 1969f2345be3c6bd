#!/bin/bash
mkdir -p gpurun_out/r03
bash tools/profile_round.sh r03b '^4s_ragged$' > gpurun_out/r03/prof9.log 2>&1
tail -2 gpurun_out/r03/prof9.log
