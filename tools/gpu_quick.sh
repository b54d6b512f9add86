#!/bin/bash
# selected GPU tests then a python tool: usage (under gpurun): tools/gpu_quick.sh <tag> "<pytest -k expr>" tools/x.py [args]
set -uo pipefail
tag=$1; sel=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$sel" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "$sel" > $out/pytest_gpu.log 2>&1 && echo "pytest ok: $(tail -1 $out/pytest_gpu.log)" || { echo "pytest FAILED"; tail -40 $out/pytest_gpu.log; exit 1; }
fi
if [ $# -gt 0 ]; then timeout -k 10 600 python3 "$@" 2>&1 | tee $out/tool.log; fi
