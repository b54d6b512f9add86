#!/bin/bash
# the common tail's share of a run in das_exact_nd_kernel's item queues (tuning build tools/ab/nd_tail.so, AWPU_ND_TAIL = percent of a run),
# alternating repetitions on one box.  usage (under gpurun): tools/gpu_nd_tail.sh <tag> "<percents>"
set -euo pipefail
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/nd_tail.so beamforming-lk_amd/libawpu_hip.so
for rep in 1 2 3; do
for t in ${2:-25 12 6 3}; do
  AWPU_ND_TAIL=$t timeout -k 10 200 python bench.py --math exact --cpu-seconds 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('tail $t%%: value %.0f kernel %.3f ms parity %.2e'%(d['value'],d['roofline']['kernel_ms'],d['parity_max_rel_err']))" | tee -a $out/tail.log
done
done
