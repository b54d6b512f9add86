#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
out=gpurun_out/r5q; mkdir -p $out
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/ndh_nw.so beamforming-lk_amd/libawpu_hip.so
for nw in 16 8 4; do
  echo "== AWPU_NDH_WAVES=$nw"
  AWPU_NDH_WAVES=$nw python tools/single_frame_rate.py --math exact c2 headline 2>/dev/null
done
echo "== default rule"
python tools/single_frame_rate.py --math exact ref_default c2 headline 2>/dev/null
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "exact_mode or reference_order or dc_biased" 2>&1 | tail -3
