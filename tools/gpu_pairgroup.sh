#!/bin/bash
# quad kernel: frame pairs an XCD works on at a time (AWPU_FAST_PAIRGROUP) on a given workload: usage (under gpurun): bash tools/gpu_pairgroup.sh <tag> <workload>
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out; wl=${2:-c5}
for rep in 1 2; do
for g in 1 2 4 8; do
  AWPU_FAST_PAIRGROUP=$g timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extras --workload $wl > $out/g${g}_$rep.json 2> $out/g${g}_$rep.err
  python - "$out/g${g}_$rep.json" $g <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("pair group %s: value %.0f frames/s  kernel %.3f ms  valu %.3f" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"]))
PY
done
done
