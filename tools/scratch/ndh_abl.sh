#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so beamforming-lk_amd/libawpu_hip.so' EXIT
for v in base noload; do
  cp tools/ab/ndt_$v.so beamforming-lk_amd/libawpu_hip.so
  echo "== $v"
  python tools/single_frame_rate.py --math exact ref_default c2 headline c3 2>/dev/null | cut -c1-110
done
