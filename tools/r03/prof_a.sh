#!/bin/bash
bash tools/profile_round.sh r03a '^(2s|4s|bl|g77|bf|n79|zq|zq_pa|zq_nz100|2s_f32|n79_f32|2s_nb107|n79_nb107|zq_nb107)$' 2>&1 | tail -5
