#!/bin/bash
# tuning: quad-kernel variants / knobs side by side on one box.  usage: gpu_var.sh <tag> "<ENV=VAL ...>" ...
set -euo pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras ${BENCH_ARGS:-} > $out/c${i}_$rep.json 2> $out/c${i}_$rep.err
    python - "$out/c${i}_$rep.json" "$cfg" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("%-44s value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
  done
done
