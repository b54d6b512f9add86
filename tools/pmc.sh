#!/bin/bash
# pmc.sh -- collect SQ counters for one bench.py launch set (counters only: no trace domains).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
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
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
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
for k in sorted(tot): print(f"{k:28s} {tot[k]/n[k]:16.0f}   (avg over {n[k]} launches)")
PY
