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
    pkg._build.generate_blocks()  # csrc/das_fast_trip.inc is generated at build time (tools/gen_trip_asm.py), not tracked
    out = tmp_path_factory.mktemp("asm") / "das_fast.s"
    subprocess.run([pkg._build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO / 'include'}", f"-I{CSRC}",
                    "-S", "--cuda-device-only", "-o", str(out), str(CSRC / "das_fast.hip")], check=True, capture_output=True)
    meta = {}
    for block in out.read_text().split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                      for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return meta


def test_generated_blocks_are_current(pkg, tmp_path):
    """The das_fast_trip.inc the library was built from is what tools/gen_trip_asm.py writes today with its default
    settings (nobody edited the generated file, no tuning variable leaked into a shipping build)."""
    import os
    import sys

    inc = pkg._build.generate_blocks()
    env = {k: v for k, v in os.environ.items() if not k.startswith(("QUAD", "TRIP_", "PAIR_DEPTH", "FIR_PRIO", "BLOCK_END_PRIO", "ND_"))}
    env["TRIP_INC_OUT"] = str(tmp_path / "fresh.inc")
    subprocess.run([sys.executable, str(REPO / "tools" / "gen_trip_asm.py")], check=True, capture_output=True, env=env)
    assert (tmp_path / "fresh.inc").read_text() == inc.read_text()
    tracked = subprocess.run(["git", "-C", str(REPO), "ls-files", "beamforming-lk_amd/csrc/das_fast_trip.inc"], capture_output=True, text=True)
    assert tracked.returncode != 0 or tracked.stdout.strip() == "", "the generated include is not to be committed"


# the shapes launch() can pick without a tuning knob (awpu_hip.cpp); stamped (diagnostic) builds and the shapes only
# tuning builds reach are not timed and may spill
PRODUCTION = [r"das_quad_kernelILb0ELi0E", r"das_quadh_kernelILi[12]ELb0E", r"das_quadh_stationary_kernelILi[12]E", r"das_pair_kernelILi4ELb0ELb1E",
              r"das_pair_stationary_kernelILb1E", r"das_fast_db_kernelILi16ELi[48]ELi\d+ELi4ELb0E",
              r"das_fast_kernelILi8ELi[24]ELi1ELi4E", r"das_fir8_plane_kernelILi0E", r"das_exact_pair_kernel", r"das_exact_quad_kernel",
              r"das_exact_nd_kernelILi[12]ELb0E", r"das_exact_ndh_kernelILi[12]ELb[01]E", r"das_exact_ndp_kernel"]


def test_sweep_kernels_do_not_spill(kernel_metadata):
    checked = 0
    for name, m in kernel_metadata.items():
        if not any(re.search(pat, name) for pat in PRODUCTION):
            continue
        checked += 1
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        assert m["vgpr_count"] <= 128, (name, m)
    assert checked >= 20, sorted(kernel_metadata)


def test_tuning_build_still_links(pkg, tmp_path):
    """The kernels, stamped instances and environment knobs that round 4 moved out of the shipping library live on behind
    -DAWPU_TUNING_BUILD / -DAWPU_TIMING_BUILD (AWPU_EXTRA_HIPCC_FLAGS): that build must keep compiling and linking, or the
    measurements DESIGN.md cites could not be repeated.  It reads the round-1..3 variables; the shipping library does not."""
    pkg._build.generate_blocks()
    out = tmp_path / "libawpu_tuning.so"
    srcs = [str(CSRC / f) for f in ("das_kernels.hip", "das_fast.hip", "awpu_hip.cpp", "geometry_host.cpp")]
    subprocess.run([pkg._build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result",
                    "-Werror=inline-asm", "-x", "hip", "-DAWPU_TIMING_BUILD", f"-I{REPO / 'include'}", f"-I{CSRC}", *srcs, "-o", str(out)],
                   check=True, capture_output=True)
    names = {ln for ln in subprocess.run(["strings", "-a", str(out)], capture_output=True, text=True, check=True).stdout.splitlines()
             if re.fullmatch(r"AWPU_[A-Z0-9_]+", ln)}
    assert {"AWPU_SHAPE", "AWPU_FAST_DEBUG", "AWPU_FAST_QUADS", "AWPU_FAST_PAIRGROUP", "AWPU_FIR8_STATIC"} <= names


def test_every_sweep_launcher_checks_its_reach():
    """Round-3 advisor: a launch whose kernel can read past a table or a packed buffer must be refused on the host.  Every sweep
    launcher of das_kernels.h takes the caller's `Extents` (what was allocated) and compares its kernel's reach with it
    (`within(...)` in das_fast.hip) before hipLaunchKernelGGL; the allocation sites use the same named prefetch constants."""
    import re
    hdr = (REPO / "beamforming-lk_amd" / "csrc" / "das_kernels.h").read_text()
    src = (REPO / "beamforming-lk_amd" / "csrc" / "das_fast.hip").read_text()
    host = (REPO / "beamforming-lk_amd" / "csrc" / "awpu_hip.cpp").read_text()
    sweeps = ["launch_das_pairs", "launch_das_pairs_stationary", "launch_das_exact_pairs", "launch_das_exact_quads", "launch_das_fir8_planes",
              "launch_das_quads", "launch_das_quadh", "launch_das_quadh_stationary", "launch_das_fast"]
    for name in sweeps:
        decl = re.search(r"hipError_t %s\(([^;]*)\);" % name, hdr)
        assert decl and "const Extents &have" in decl.group(1), name
        for call in re.findall(r"awpu::%s\(([^;]*)\);" % name, host):
            assert "{" in call or "have" in call, (name, call)  # the host passes what it allocated
    assert src.count("within({") >= len(sweeps)
    for const in ("kPairTablePrefetch", "kQuadTablePrefetch", "kFir8PlaneTablePrefetch"):
        assert const in hdr and const in src
