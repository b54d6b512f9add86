#!/bin/bash
# rocprofv3 evidence for one bench.py command line: kernel stats, SQ counters, HBM-side counters (separate passes)
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -euo pipefail
# One process only: under rocprofv3 the GPU is initialised before bench.py's main() runs, and `--gpus N` would start child
# ranks from that process (the exec this pool forbids).  Profile one rank's slab instead: `--workload c5`.
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: --gpus is not allowed under the profiler; use --workload c5 (one rank's slab)" >&2; exit 2;; esac; done
: "${GRAFT_REPO_ROOT:?run under gpurun}"
tag=$1; shift
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
export AWPU_NO_BUILD=1
export AWPU_UNDER_PROFILER=1  # bench.py refuses to start child ranks when it sees this (not AWPU_NO_BUILD, which only stops rebuilds)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-seconds 0 --no-extras "$@" > $out/bench_under_rocprof.json 2> $out/stats.log
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -6 $out/kernel_stats.csv
# the same launches with the first (code-load) one and the bench's warm-up steps dropped: what a per-step figure may cite
skip=4; prev=""; for a in "$@"; do [ "$prev" = "--warmup" ] && skip=$((a + 1)); prev=$a; done
python3 tools/warm_kernel_stats.py $out/stats $skip > $out/kernel_stats_warm.csv
head -4 $out/kernel_stats_warm.csv
bash tools/pmc.sh $tag/pmc --no-extras "$@" > $out/pmc_sq_summary.txt
bash tools/pmc_hbm.sh $tag/hbm --no-extras "$@" > $out/pmc_hbm_summary.txt
cat $out/pmc_sq_summary.txt $out/pmc_hbm_summary.txt
