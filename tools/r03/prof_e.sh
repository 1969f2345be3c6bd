#!/bin/bash
bash tools/profile_round.sh r03e '^(zq|zq_nz100|zq_pa|zq_pa_nz100|zq_pa_nb107|zq_pa_nb38|zq_pa_ragged|zq_nb107|zq_nb38_nz100|zq_integrated|zq_ragged|zq_nz100_ragged|band_zq|band_cfg4|zq_nb12|zq_nb8_wave)$' 2>&1 | tail -3
