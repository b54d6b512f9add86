#!/bin/bash
# does the headline step time depend on the number of warm-up steps?  usage (under gpurun): bash tools/gpu_warm.sh <tag>
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2; do
for w in 3 20 60; do
  timeout -k 10 300 python bench.py --cpu-seconds 0 --no-extras --warmup $w > $out/w${w}_$rep.json 2> $out/w${w}_$rep.err
  python - "$out/w${w}_$rep.json" $w <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("warmup %-3s ms_per_step %.3f kernel %.3f value %.0f" % (sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"]))
PY
done
done
timeout -k 10 300 python bench.py --cpu-seconds 0 > $out/full.json 2> $out/full.err
python - $out/full.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
for w in d["workloads"]: print(w["workload"][:30], w["steps"], w["kernel_ms"], w["valu_frac"])
print(d["projected_scaling"]["ceiling_x"], d["single_frame"]["valu_frac"])
PY
