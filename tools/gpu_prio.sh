#!/bin/bash
# stream priorities of the N > 1 step loop (sweep, aux) through projected_scaling on one GPU: usage (under gpurun): bash tools/gpu_prio.sh <tag>
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2; do
for p in "-1,0" "0,-1" "0,0"; do
  BENCH_STREAM_PRIO="$p" timeout -k 10 300 python bench.py --cpu-seconds 0 > $out/p_${p}_$rep.json 2> $out/p_${p}_$rep.err
  python - "$out/p_${p}_$rep.json" "$p" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
p=d["projected_scaling"]
print("prio %-6s one-gpu %.3f ms  slab kernel %.3f (bare %.3f)  step wall %.3f  peer wall %.3f  ceiling %.2f" % (sys.argv[2], d["ms_per_step"], p["slab_kernel_ms"], p["slab_kernel_ms_bare"], p["step_wall_ms"], p["peer_step_wall_ms"], p["ceiling_x"]))
PY
done
done
