#!/bin/bash
# FIR8 plane kernel: static-pitch block against the generic one (same library, AWPU_FIR8_STATIC): usage (under gpurun): bash tools/gpu_fir_static.sh <tag>
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out
python -m pytest tests -m gpu -x -q -k "fir" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for rep in 1 2; do
for v in 0 1; do
  AWPU_FIR8_STATIC=$v timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extras --workload c3 --interp fir8 --steps 5 --warmup 2 > $out/s${v}_$rep.json 2> $out/s${v}_$rep.err
  python - "$out/s${v}_$rep.json" $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("static=%s value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
