#!/bin/bash
# per-workgroup timeline of das_exact_nd_kernel from a tuning build (tools/ab/<lib>): usage (under gpurun) tools/gpu_nd_diag.sh <tag> <lib.so> [bench args]
set -uo pipefail
tag=$1; lib=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
cp beamforming-lk_amd/libawpu_hip.so $out/keep.so
trap 'cp "$out/keep.so" beamforming-lk_amd/libawpu_hip.so' EXIT
cp tools/ab/$lib beamforming-lk_amd/libawpu_hip.so
AWPU_FAST_DEBUG=16 timeout -k 10 200 python bench.py --math exact --cpu-seconds 0 --no-extras --steps 3 --warmup 1 "$@" > $out/diag.json 2> $out/diag.err
grep 'awpu diag' $out/diag.err | tail -4
