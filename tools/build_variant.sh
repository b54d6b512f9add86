#!/bin/bash
# a tuning/timing variant of the library into tools/ab/<name>.so: the generator's environment variables (QUAD1_NOBRANCH=1 ...)
# and AWPU_EXTRA_HIPCC_FLAGS apply; the tracked build's generated include is restored afterwards.
# usage: [GENERATOR VARS] tools/build_variant.sh <name>
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1
mkdir -p tools/ab
python tools/gen_trip_asm.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-result -Werror=inline-asm -x hip \
  -Iinclude -Ibeamforming-lk_amd/csrc ${AWPU_EXTRA_HIPCC_FLAGS:-} \
  beamforming-lk_amd/csrc/das_kernels.hip beamforming-lk_amd/csrc/das_fast.hip beamforming-lk_amd/csrc/awpu_hip.cpp beamforming-lk_amd/csrc/geometry_host.cpp \
  -o tools/ab/$name.so
env -i PATH="$PATH" HOME="$HOME" python tools/gen_trip_asm.py > /dev/null   # back to the defaults
touch -r beamforming-lk_amd/libawpu_hip.so beamforming-lk_amd/csrc/das_fast_trip.inc 2>/dev/null || true
echo "built tools/ab/$name.so"
