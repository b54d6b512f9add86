#!/bin/bash
# selected GPU tests, then bench.py with the given arguments: usage (under gpurun): tools/gpu_bench.sh <tag> "<pytest -k expr or ''>" [bench args...]
set -uo pipefail
tag=$1; sel=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$sel" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "$sel" > $out/pytest_gpu.log 2>&1 && echo "pytest ok: $(tail -1 $out/pytest_gpu.log)" || { echo "pytest FAILED"; tail -40 $out/pytest_gpu.log; exit 1; }
fi
timeout -k 10 600 python bench.py "$@" > $out/bench.json 2> $out/bench.err && echo "bench ok" || { echo "bench FAILED"; tail -30 $out/bench.err; exit 1; }
python3 - $out/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.0f frames/s  ms/step %.3f  kernel %s %.3f ms (min %.3f median %.3f)  valu %.3f  parity %.2e ok=%s" % (
    d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"].get("kernel_ms_min", 0),
    d["roofline"].get("kernel_ms_median", 0), d["valu"]["frac"], d["parity_max_rel_err"], d["parity"]["ok"]))
if "single_frame" in d:
    s = d["single_frame"]; print("single_frame: %s %.1f us/frame valu %.3f parity %.2e" % (s.get("kernel"), s["ms_per_frame_device"] * 1e3, s["valu_frac"], s["parity_max_rel_err"]))
if "reference_default" in d:
    r = d["reference_default"]; print("reference_default: %s %.1f us/frame device, %.1f us host call, parity %.2e ok=%s" % (r.get("kernel"), r["ms_per_frame_device"] * 1e3, r["ms_per_host_call"] * 1e3, r["parity_max_rel_unfloored"], r["parity_ok"]))
for w in d.get("workloads", []):
    print("workload %-40s %s %8.0f frames/s kernel %.3f ms valu %.3f parity %.2e ok=%s" % (w["workload"][:40], w.get("kernel"), w["value"], w["kernel_ms"], w["valu_frac"], w["parity_max_rel_unfloored"], w["parity_ok"]))
if "parity_dc" in d:
    for c in d["parity_dc"]["cases"]: print("parity_dc", c["golden"], "exact", ["%.1e" % v for v in c["exact"]], "fast", ["%.1e" % v for v in c["fast"]])
if "projected_scaling" in d:
    p = d["projected_scaling"]; print("projected ceiling_x %.2f (raw_scatter %s)" % (p["ceiling_x"], p.get("raw_scatter", {}).get("ceiling_x")))
PY
