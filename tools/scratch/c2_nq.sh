#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
for shape in default exact_nd1 exact_nd2; do
  for wl in c2 c3; do
    if [ "$shape" = default ]; then unset AWPU_SHAPE; else export AWPU_SHAPE=$shape; fi
    timeout -k 10 300 python bench.py --workload $wl --cpu-seconds 0 --no-extras --steps 12 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$shape $wl', round(d['value']), 'frames/s kernel', round(d['roofline']['kernel_ms'],4), 'ms', d['roofline']['kernel'], d['parity_max_rel_err'])"
  done
done
unset AWPU_SHAPE
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "single_frames_are_the_reference_bits" 2>&1 | tail -3
