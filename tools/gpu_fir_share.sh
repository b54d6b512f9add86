#!/bin/bash
# FIR8 plane kernel: vertical pixel quads with shared samples against consecutive pixels (same library, AWPU_FIR8_SHARE): usage (under gpurun): bash tools/gpu_fir_share.sh <tag> [workload]
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out; wl=${2:-c3}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "fir" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for rep in 1 2; do
for v in 0 1; do
  AWPU_FIR8_SHARE=$v timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extras --workload $wl --interp fir8 --steps 5 --warmup 2 > $out/s${v}_$rep.json 2> $out/s${v}_$rep.err
  python - "$out/s${v}_$rep.json" $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("share=%s value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.3e" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
