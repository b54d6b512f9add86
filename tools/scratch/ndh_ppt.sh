#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so beamforming-lk_amd/libawpu_hip.so' EXIT
for rep in 1 2; do
for v in 1 2 5; do
  cp tools/ab/ndh_ppt$v.so beamforming-lk_amd/libawpu_hip.so
  echo "== pieces per trip x$v"
  python tools/single_frame_rate.py --math exact c2 headline c3 2>/dev/null
done
done
