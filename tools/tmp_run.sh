#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "exact or dc_biased" > gpurun_out/r04b/pytest.log 2>&1 && echo "pytest ok: $(tail -1 gpurun_out/r04b/pytest.log)" || { tail -30 gpurun_out/r04b/pytest.log; exit 1; }
bash tools/gpu_profile.sh r04b_headline_exact --math exact --steps 8 --warmup 3 > gpurun_out/r04b/profile_exact.log 2>&1 && echo profile ok
head -3 gpurun_out/r04b_headline_exact/kernel_stats_warm.csv | cut -c1-160
tail -2 gpurun_out/r04b_headline_exact/pmc_hbm_summary.txt
