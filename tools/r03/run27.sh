#!/bin/bash
mkdir -p gpurun_out/r03/translation
rm -f gpurun_out/r03/translation/summary.txt
bash tools/translation_counters.sh $PWD/gpurun_out/r03/translation 2>&1 | tail -20
