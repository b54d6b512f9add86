#!/bin/bash
# round 4, first GPU pass: the new reference-order kernel's tests, then its rate at the headline shape
set -uo pipefail
out=gpurun_out/r4a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s -k "exact or dc or golden" > $out/pytest_new.log 2>&1 && echo "new tests ok" || { echo "new tests FAILED"; tail -40 $out/pytest_new.log; exit 1; }
timeout -k 10 200 python bench.py --math exact --no-extras --cpu-seconds 0 --steps 5 --warmup 2 > $out/bench_exact.json 2> $out/bench_exact.err && echo "bench exact ok" || { tail -20 $out/bench_exact.err; exit 1; }
AWPU_EXACT_PAIRS=0 timeout -k 10 300 python bench.py --math exact --no-extras --cpu-seconds 0 --steps 2 --warmup 1 > $out/bench_exact_old.json 2> $out/bench_exact_old.err && echo "bench exact old ok"
tail -c 1500 $out/bench_exact.json; echo; tail -c 600 $out/bench_exact_old.json
