#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/libA.so vs tools/ab/libB.so (bench legs only).
# usage (under gpurun): BENCH_ARGS="..." bash tools/gpu_ab_lib.sh <tag>
set -euo pipefail
out=gpurun_out/${1:-ablib}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
for rep in 1 2 3; do
for v in A B; do
  cp tools/ab/lib$v.so beamforming-lk_amd/libawpu_hip.so
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras ${BENCH_ARGS:-} > $out/${v}_$rep.json 2> $out/${v}_$rep.err
  python - "$out/${v}_$rep.json" $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("lib", sys.argv[2], "value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
cp tools/ab/libB.so beamforming-lk_amd/libawpu_hip.so
