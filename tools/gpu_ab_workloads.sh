#!/bin/bash
# the default bench line (all workloads) of several library builds side by side on one box (under gpurun):
#   tools/gpu_ab_workloads.sh <tag> a.so b.so ...
set -euo pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT  # whatever happens below, the shipping library comes back
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  cp tools/ab/$v beamforming-lk_amd/libawpu_hip.so
  timeout -k 10 300 python bench.py --cpu-seconds 0 > $out/${v}_$rep.json 2> $out/${v}_$rep.err
  python - "$out/${v}_$rep.json" $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("%-10s headline %.3f ms" % (sys.argv[2], d["roofline"]["kernel_ms"]), " | ".join("%s %.3f" % (w["workload"].split(":")[0], w["kernel_ms"]) for w in d.get("workloads", [])),
      "| single %.1f us | ref_default %.1f us" % (d["single_frame"]["ms_per_frame_device"] * 1e3, d["reference_default"]["ms_per_frame_device"] * 1e3))
PY
done
done
