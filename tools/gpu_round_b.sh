#!/bin/bash
# measurement pass of round 2: every BASELINE workload through bench.py, then the profiles (run under gpurun)
set -euo pipefail
out=gpurun_out/r02b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
for wl in headline c1 c2 c3 c4 c5; do
  timeout -k 10 400 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "bench $wl done"
done
timeout -k 10 400 python bench.py --workload c3 --interp fir8 > $out/bench_c3_fir8.json 2> $out/bench_c3_fir8.err
BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 > $out/bench_rehearsal2.json 2> $out/bench_rehearsal2.err
bash tools/gpu_profile.sh r02_headline_b128 > $out/profile_b128.log 2>&1
bash tools/gpu_profile.sh r02_headline_b1 --batch 1 --steps 200 --warmup 20 > $out/profile_b1.log 2>&1
echo all done
