"""The sweep kernels keep every accumulator in registers: the hand-scheduled asm blocks pin or hard-code VGPRs
(tools/gen_trip_asm.py), so a register that the compiler has to spill is silent slowness -- or, around an inline-asm
load, a wrong result.  Compile the kernel file to assembly for gfx950 (hipcc cross-compiles without a GPU) and read
the metadata of every sweep kernel: no spills, no scratch, at most 128 registers (4 waves per SIMD)."""
import re
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "beamforming-lk_amd" / "csrc"


@pytest.fixture(scope="module")
def kernel_metadata(tmp_path_factory, pkg):
    out = tmp_path_factory.mktemp("asm") / "das_fast.s"
    subprocess.run([pkg._build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO / 'include'}", f"-I{CSRC}",
                    "-S", "--cuda-device-only", "-o", str(out), str(CSRC / "das_fast.hip")], check=True, capture_output=True)
    meta = {}
    for block in out.read_text().split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                      for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return meta


def test_generated_blocks_are_current():
    """das_fast_trip.inc is what tools/gen_trip_asm.py writes (nobody edited the generated file, or forgot to
    regenerate it after changing the generator)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("gen_trip_asm", REPO / "tools" / "gen_trip_asm.py")
    before = (CSRC / "das_fast_trip.inc").read_text()
    env_backup = {k: os.environ.pop(k) for k in ("QUAD_VARIANTS", "TRIP_PRIO", "TRIP_DEPTH", "PAIR_DEPTH", "QUAD_CHAIN", "QUAD_YMAP", "QUAD_XMAP", "QUAD1_TIMING_SKIP", "QUAD1_NOBRANCH", "QUAD1_EARLY_X", "FIR_PRIO", "QUAD_PRIO_COARSE", "QUAD1_PRIO_COARSE")
                  if k in os.environ}
    try:
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.main()
        assert (CSRC / "das_fast_trip.inc").read_text() == before
    finally:
        (CSRC / "das_fast_trip.inc").write_text(before)
        os.environ.update(env_backup)


# the shapes launch() can pick without a tuning knob (awpu_hip.cpp); stamped (diagnostic) builds and the shapes only
# AWPU_FAST_VARIANT reaches are not timed and may spill
PRODUCTION = [r"das_quad_kernelILb0ELi0E", r"das_quad1_kernelILi[12]ELb0E", r"das_quadh_kernelILi[12]ELb0E", r"das_pair_kernelILi4ELb0ELb[01]E",
              r"das_pair_stationary_kernelILb[01]E", r"das_fast_db_kernelILi16ELi[48]ELi\d+ELi4ELb0E",
              r"das_fast_kernelILi8ELi[24]ELi1ELi4E", r"das_fir8_plane_kernelILi0E"]


def test_sweep_kernels_do_not_spill(kernel_metadata):
    checked = 0
    for name, m in kernel_metadata.items():
        if not any(re.search(pat, name) for pat in PRODUCTION):
            continue
        checked += 1
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        assert m["vgpr_count"] <= 128, (name, m)
    assert checked >= 12, sorted(kernel_metadata)
