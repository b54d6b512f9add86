#!/bin/bash
# Round 5's evidence pass: ONE gpurun call, ONE box, every mode (run under gpurun from the repo root).  Per configuration: the bench line
# (NOT under the profiler) first, then rocprofv3 kernel stats (incl. the warm-only table), SQ counters and HBM-side counters (separate
# passes, counters only) of the same command line -- so every kernel time in profiles/r05_* sits beside the ms_per_step of its own box.
# usage: tools/gpu_round5.sh <tag> [configs: default fast batch1 c3fir8 c5 | all]
set -uo pipefail
tag=${1:-r05}; shift || true
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] || [ "${cfgs[0]}" = all ] && cfgs=(default fast batch1 c3fir8 c5)
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
export AWPU_NO_BUILD=1
declare -A ARGS=( [default]="" [fast]="--math fast" [batch1]="--batch 1 --steps 200 --warmup 20" \
                  [c3fir8]="--workload c3 --interp fir8 --math fast --steps 4 --warmup 2" [c5]="--workload c5 --steps 4 --warmup 2" )
for c in "${cfgs[@]}"; do
  a=${ARGS[$c]}
  d=$out/$c; mkdir -p $d
  if [ "$c" = default ]; then  # the full default line, as the driver runs it
    timeout -k 10 400 python bench.py > $d/bench.json 2> $d/bench.err || { echo "bench $c failed"; tail -5 $d/bench.err; exit 1; }
  else
    timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extras $a > $d/bench.json 2> $d/bench.err || { echo "bench $c failed"; tail -5 $d/bench.err; exit 1; }
  fi
  echo "bench $c done"
  export AWPU_UNDER_PROFILER=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/stats -- python3 bench.py --cpu-seconds 0 --no-extras $a > $d/bench_under_rocprof.json 2> $d/stats.log || { echo "stats $c failed"; tail -5 $d/stats.log; exit 1; }
  find $d/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $d/kernel_stats.csv
  skip=4; prev=""; for w in $a; do [ "$prev" = "--warmup" ] && skip=$((w + 1)); prev=$w; done
  python3 tools/warm_kernel_stats.py $d/stats $skip > $d/kernel_stats_warm.csv
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    g=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --output-format csv -d $d/pmc/$g -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras $a > $d/pmc_$g.log 2>&1 || { echo "pmc $g $c failed"; tail -5 $d/pmc_$g.log; exit 1; }
  done
  unset AWPU_UNDER_PROFILER
  python3 - $d <<'PY' > $d/pmc_summary.txt
import csv, glob, sys, collections
d = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter(); names = collections.Counter()
for f in glob.glob(d + "/pmc/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "das_" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1; names[row["Kernel_Name"][:60]] += 1
print("# kernels:", dict(names))
for k in sorted(tot): print(f"{k:28s} {tot[k]/n[k]:18.1f}   (avg over {n[k]} launches)")
if "TCC_MISS_sum" in tot and "WRITE_SIZE" in tot:
    miss, wr, fetch = tot["TCC_MISS_sum"] / n["TCC_MISS_sum"], tot["WRITE_SIZE"] / n["WRITE_SIZE"], tot["FETCH_SIZE"] / n["FETCH_SIZE"]
    print(f"# traffic = 128 B x TCC_MISS_sum + 1024 B x WRITE_SIZE = {128 * miss + 1024 * wr:.0f} bytes per launch (2 x FETCH_SIZE + WRITE_SIZE: {2048 * fetch + 1024 * wr:.0f})")
PY
  echo "profile $c done"
done
echo all done
