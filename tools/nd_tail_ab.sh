#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/nd_tail.so beamforming-lk_amd/libawpu_hip.so
for rep in 1 2 3; do
for t in 100 25 12 6; do
  AWPU_ND_TAIL=$t timeout -k 10 200 python bench.py --math exact --cpu-seconds 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('tail $t%%: value %.0f kernel %.3f ms parity %.2e'%(d['value'],d['roofline']['kernel_ms'],d['parity_max_rel_err']))"
done
done
AWPU_ND_TAIL=12 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "exact_mode or packed or c4_rank or c5_" 2>&1 | tail -2
