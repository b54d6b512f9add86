#!/bin/bash
# pmc_hbm.sh -- HBM-side traffic of the sweep launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2), counters only, no trace domains.
# usage: tools/pmc_hbm.sh <outdir-under-gpurun_out> [bench args...]
set -euo pipefail
# One process only: under rocprofv3 the GPU is initialised before bench.py's main() runs, and `--gpus N` would start child
# ranks from that process (the exec this pool forbids).  Profile one rank's slab instead: `--workload c5`.
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "$0: --gpus is not allowed under the profiler; use --workload c5 (one rank's slab)" >&2; exit 2;; esac; done
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
out=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
# Build BEFORE the profiler is in the picture: under rocprofv3 every child inherits the preloaded tool
# library, and a compiler or make started from a process whose GPU it has initialised is an exec the pool
# forbids.  AWPU_NO_BUILD=1 then makes the binding and the oracle loader refuse to build (they raise).
if [ "${AWPU_NO_BUILD:-}" != "1" ]; then  # (a caller that has built already says so)
  python3 -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1
fi
export AWPU_NO_BUILD=1
export AWPU_UNDER_PROFILER=1  # bench.py refuses to start child ranks when it sees this (not AWPU_NO_BUILD, which only stops rebuilds)
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/$tag -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $out/$tag.log 2>&1 || { echo "rocprofv3 pass $tag failed, see $out/$tag.log" >&2; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "das_" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]/n[k]:18.1f}   (avg over {n[k]} launches)")
# calibrated on the kernels' own access patterns (profiles/r03_fetch_calibration.txt): one missed 128-byte line = one TCC_MISS
if "TCC_MISS_sum" in tot and "WRITE_SIZE" in tot:
    miss, wr, fetch = tot["TCC_MISS_sum"] / n["TCC_MISS_sum"], tot["WRITE_SIZE"] / n["WRITE_SIZE"], tot["FETCH_SIZE"] / n["FETCH_SIZE"]
    print(f"# traffic = 128 B x TCC_MISS_sum + 1024 B x WRITE_SIZE = {128 * miss + 1024 * wr:.0f} bytes per launch "
          f"(2 x FETCH_SIZE + WRITE_SIZE, the correction for pure 16-byte-per-lane streams: {2048 * fetch + 1024 * wr:.0f})")
PY
