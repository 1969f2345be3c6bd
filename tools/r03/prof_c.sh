#!/bin/bash
bash tools/profile_round.sh r03c '^(zq_pa_nb38|zq_pa_nz100|n79_nz100|zq_pa_nb107|.*_ragged)$' 2>&1 | tail -5
