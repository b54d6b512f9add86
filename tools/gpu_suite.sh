#!/bin/bash
# the whole -m gpu suite + the C++ host tests, log under gpurun_out/<tag>/ (run from the repo root under gpurun)
# usage: tools/gpu_suite.sh <tag> [pytest -k expression]
set -uo pipefail
tag=${1:-suite}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "${2:-}" ]; then sel=(-k "$2"); else sel=(); fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s "${sel[@]}" > $out/pytest_gpu.log 2>&1 && echo "pytest ok" || { echo "pytest FAILED"; tail -40 $out/pytest_gpu.log; exit 1; }
tail -3 $out/pytest_gpu.log
if [ -x tests/host/test_mimo_worker ] && [ -z "${2:-}" ]; then
  timeout -k 10 120 tests/host/test_mimo_worker > $out/host.log 2>&1 && echo "host ok" || { echo "host FAILED"; tail -20 $out/host.log; exit 1; }
fi
