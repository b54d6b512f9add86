#!/bin/bash
# single-frame rate of several library builds side by side on one box (under gpurun): tools/gpu_ab_single.sh <tag> "<workloads>" a.so b.so ...
set -euo pipefail
out=gpurun_out/$1; shift
wl=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT  # whatever happens below, the shipping library comes back
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  cp tools/ab/$v beamforming-lk_amd/libawpu_hip.so
  echo "== $v rep $rep" | tee -a $out/single.log
  timeout -k 10 200 python tools/single_frame_rate.py $wl 2>&1 | tee -a $out/single.log
done
done
