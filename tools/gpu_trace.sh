#!/bin/bash
# rocprofv3 kernel trace of a python tool (one process): usage (under gpurun): tools/gpu_trace.sh <tag> <skip> tools/x.py [args]
set -uo pipefail
tag=$1; skip=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
export AWPU_NO_BUILD=1 AWPU_UNDER_PROFILER=1
python3 "$@" > $out/plain.log 2>&1; cat $out/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/under_rocprof.log 2> $out/rocprof.err || { tail -20 $out/rocprof.err; exit 1; }
cat $out/under_rocprof.log
python3 tools/warm_kernel_stats.py $out/trace $skip > $out/kernel_stats_warm.csv
cat $out/kernel_stats_warm.csv
python3 - $out/trace <<'PY'
import csv, glob, sys, statistics, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(n0[:40], n1[:40])].append(s1 - e0)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
    print("gap end->start ns", k, "n", len(v), "median", statistics.median(v))
PY
