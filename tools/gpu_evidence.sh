#!/bin/bash
# the slower evidence of a round (run under gpurun, after tools/gpu_round.sh): the single-frame kernel trace, the full-grid
# parity survey of every shape, the randomised soak.  usage: tools/gpu_evidence.sh <tag> [seeds...]
set -uo pipefail
tag=$1; shift
out=gpurun_out/${tag}_evidence
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
bash tools/gpu_trace.sh ${tag}_trace 21 tools/single_frame_rate.py ref_default c1 c2 headline > $out/trace_single_frame.txt 2>&1 && echo "trace done"
timeout -k 10 500 python3 tools/parity_survey.py --out $out/parity_survey.json > $out/parity_survey.log 2>&1 && echo "survey ok: $(tail -1 $out/parity_survey.log | cut -c1-200)" || { echo "survey FAILED"; tail -5 $out/parity_survey.log; }
SEEDS="${*:-11 12}" timeout -k 10 900 bash tools/gpu_soak.sh > $out/random_soak.log 2>&1 && echo "soak ok: $(grep -c '^ok' $out/random_soak.log) runs" || { echo "soak FAILED"; grep -v '^ok' $out/random_soak.log | tail -5; }
