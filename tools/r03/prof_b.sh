#!/bin/bash
bash tools/profile_round.sh r03b '^(zq_nb38_nz100|2s_nb38|zq_nb12|2s_nb12|zq_nb8_wave|2s_integrated|n79_integrated|zq_integrated|band_zq|2s_125k|band_cfg4|epilogue|epilogue_nb38)$' 2>&1 | tail -5
