#!/bin/bash
# where does one pixel per wave stop paying?  grids of more than 16 pixels per CU (a second round of workgroups) against the quad kernel's
# rule, tuning build tools/ab/ndp.so.  usage (under gpurun): tools/gpu_ndp_rule.sh <tag>
set -euo pipefail
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/ndp.so beamforming-lk_amd/libawpu_hip.so
for w in default 1; do
  echo "== AWPU_NDH_WAVES=$w" | tee -a $out/rule.log
  if [ $w = default ]; then unset AWPU_NDH_WAVES; else export AWPU_NDH_WAVES=$w; fi
  timeout -k 10 300 python tools/single_frame_rate.py --math exact ${WL:-c2@72 c2@80 c2@88 c2@96 ref_4arrays c1} 2>&1 | grep -v amdgpu.ids | tee -a $out/rule.log
done
