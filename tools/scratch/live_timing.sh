#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/live_timing.so beamforming-lk_amd/libawpu_hip.so
AWPU_LIVE_TIMING=1 python3 tools/scratch/host_call_ref.py 2>&1 | grep -v amdgpu.ids
echo "-- spin on hipStreamQuery"
AWPU_LIVE_SPIN=1 AWPU_LIVE_TIMING=1 python3 tools/scratch/host_call_ref.py 2>&1 | grep -v amdgpu.ids
