#!/bin/bash
# several builds of the library side by side on one box: usage (under gpurun): BENCH_ARGS=".." bash tools/gpu_ab_libs.sh <tag> libX.so libY.so ...
set -euo pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT  # whatever happens below, the shipping library comes back
for rep in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
  cp tools/ab/$v beamforming-lk_amd/libawpu_hip.so
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras ${BENCH_ARGS:-} > $out/${v}_$rep.json 2> $out/${v}_$rep.err
  python - "$out/${v}_$rep.json" $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("%-12s value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
