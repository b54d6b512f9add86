#!/bin/bash
# A/B of the batched sweep on one box: quad shape vs pair shape (bench legs only, no CPU baseline)
set -euo pipefail
out=gpurun_out/${1:-ab}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for q in 0 1; do
  AWPU_FAST_QUADS=$q timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras ${BENCH_ARGS:-} > $out/q${q}_$rep.json 2> $out/q${q}_$rep.err
  python - "$out/q${q}_$rep.json" $q <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("quads", sys.argv[2], "value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
