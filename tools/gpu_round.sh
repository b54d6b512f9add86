#!/bin/bash
# measurement pass of a round (run under gpurun): rocprofv3 evidence for the default bench (kernel stats incl. the warm-only
# table, SQ counters, HBM-side counters in separate passes), the single-frame regime, the reference-order mode, then every
# BASELINE workload through bench.py.  usage: tools/gpu_round.sh <tag> [profiles|workloads|all]   (two gpurun calls fit the 20-minute limit)
set -euo pipefail
tag=${1:-r04}
part=${2:-all}
out=gpurun_out/${tag}_round
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
if [ "$part" != "workloads" ]; then
bash tools/gpu_profile.sh ${tag}_headline_b128 > $out/profile_b128.log 2>&1
echo "profile b128 done"
bash tools/gpu_profile.sh ${tag}_headline_b1 --batch 1 --steps 200 --warmup 20 > $out/profile_b1.log 2>&1
echo "profile b1 done"
bash tools/gpu_profile.sh ${tag}_headline_exact --math exact --steps 8 --warmup 3 > $out/profile_exact.log 2>&1
echo "profile exact done"
fi
if [ "$part" != "profiles" ]; then
for wl in headline c1 c2 c3 c4 c5; do
  timeout -k 10 500 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "bench $wl done"
done
timeout -k 10 400 python bench.py --workload c3 --interp fir8 > $out/bench_c3_fir8.json 2> $out/bench_c3_fir8.err
timeout -k 10 400 python bench.py --math exact --no-extras > $out/bench_headline_exact.json 2> $out/bench_headline_exact.err
BENCH_REHEARSAL=1 BENCH_GATHER_CHECK=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > $out/bench_rehearsal2.json 2> $out/bench_rehearsal2.err
fi
echo all done
