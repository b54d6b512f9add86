#!/bin/bash
# das_exact_ndp_kernel (one pixel per wave) against the quad kernel's 4-wave workgroups on c2, one frame per call, and what the new
# kernel costs without one of its parts (tools/ab/ndp*.so: tuning builds, tools/build_variant.sh).  usage (under gpurun): tools/gpu_ndp_ab.sh <tag>
set -euo pipefail
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT  # whatever happens below, the shipping library comes back
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "single_frames_are_the_reference_bits" 2>&1 | tail -5 | tee $out/pytest.log
cp tools/ab/ndp.so beamforming-lk_amd/libawpu_hip.so
for rep in 1 2; do
  for w in 4 1; do
    echo "== AWPU_NDH_WAVES=$w (4: quads in 4-wave workgroups, 1: one pixel per wave) rep $rep" | tee -a $out/single.log
    AWPU_NDH_WAVES=$w timeout -k 10 200 python tools/single_frame_rate.py --math exact c2 2>&1 | grep -v amdgpu.ids | tee -a $out/single.log
  done
done
for v in ndp_novalu ndp_nolds ndp_nodma ndp_nobarrier; do
  cp tools/ab/$v.so beamforming-lk_amd/libawpu_hip.so
  echo "== $v" | tee -a $out/single.log
  AWPU_NDH_WAVES=1 timeout -k 10 200 python tools/single_frame_rate.py --math exact c2 2>&1 | grep -v amdgpu.ids | tee -a $out/single.log
done
