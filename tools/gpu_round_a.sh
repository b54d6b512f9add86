#!/bin/bash
# first GPU pass of round 2: tests, host test, bench legs (run from the repo root under gpurun)
set -euo pipefail
out=gpurun_out/r02a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $out/pytest_gpu.log 2>&1 && echo "pytest ok" || { echo "pytest FAILED"; tail -30 $out/pytest_gpu.log; exit 1; }
timeout -k 10 120 tests/host/test_mimo_worker > $out/host.log 2>&1 && echo "host ok" || { echo "host FAILED"; tail -20 $out/host.log; exit 1; }
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err && echo "bench ok"
timeout -k 10 300 python bench.py --workload c5 > $out/bench_c5.json 2> $out/bench_c5.err && echo "c5 ok"
timeout -k 10 300 python bench.py --workload c3 --interp fir8 --batch 16 --steps 5 --warmup 1 --cpu-seconds 0 > $out/bench_c3_fir8.json 2> $out/bench_c3_fir8.err && echo "fir8 ok"
BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 --workload c2 > $out/bench_rehearsal2.json 2> $out/bench_rehearsal2.err && echo "rehearsal ok"
tail -c 600 $out/bench_default.json
