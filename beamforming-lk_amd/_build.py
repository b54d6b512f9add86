"""Build recipe for libawpu_hip.so (gfx950 only, in-tree).

hipcc cross-compiles without a GPU, so this runs in the authoring container and on the GPU
box alike.  The library is written next to this file; it is git-ignored but travels with
the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libawpu_hip.so"

SOURCES = [CSRC / "das_kernels.hip", CSRC / "das_fast.hip", CSRC / "awpu_hip.cpp", CSRC / "geometry_host.cpp"]
# das_fast_trip.inc -- the hand-scheduled inner loops of das_fast.hip -- is GENERATED at build time by tools/gen_trip_asm.py
# (not tracked: ~19 000 lines of asm text whose source is the generator)
GENERATOR = REPO / "tools" / "gen_trip_asm.py"
TRIP_INC = CSRC / "das_fast_trip.inc"
HEADERS = [CSRC / "das_kernels.h", GENERATOR, REPO / "include" / "awpu_hip.h"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libawpu_hip.so cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not LIB_PATH.exists():
        return True
    built = LIB_PATH.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP kernels and the C ABI into beamforming-lk_amd/libawpu_hip.so."""
    if not force and not stale():
        return LIB_PATH
    # several ranks of one job may get here at once (torch.distributed.run starts them together): one builds,
    # the others wait for the lock and then find the library fresh
    import fcntl

    with open(PKG_DIR / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():
            return LIB_PATH
        return _compile(verbose, force)


def generate_blocks(force: bool = False) -> Path:
    """Write csrc/das_fast_trip.inc with the generator's defaults (its tuning variables are for tuning builds, by hand)."""
    if not force and TRIP_INC.exists() and TRIP_INC.stat().st_mtime >= GENERATOR.stat().st_mtime:
        return TRIP_INC
    import sys

    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("QUAD", "TRIP_", "PAIR_DEPTH", "FIR_PRIO", "BLOCK_END_PRIO", "ND_"))}  # a shipping build: the defaults
    proc = subprocess.run([sys.executable, str(GENERATOR)], capture_output=True, text=True, env=env)
    if proc.returncode != 0 or not TRIP_INC.exists():
        raise RuntimeError(f"tools/gen_trip_asm.py failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return TRIP_INC


def _compile(verbose: bool, force: bool = False) -> Path:
    generate_blocks(force)
    tmp = LIB_PATH.with_suffix(f".so.tmp{os.getpid()}")
    cmd = [
        hipcc_path(),
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        "-Wall",
        "-Wno-unused-result",
        "-Werror=inline-asm",  # e.g. a reserved register on a hand-written block's clobber list: undefined behaviour, not a warning
        "-x", "hip",
        f"-I{REPO / 'include'}",
        f"-I{CSRC}",
        *os.environ.get("AWPU_EXTRA_HIPCC_FLAGS", "").split(),  # tuning builds (e.g. -DAWPU_QUAD_VARIANTS)
        *[str(s) for s in SOURCES],
        "-o", str(tmp),
    ]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    if verbose and proc.stderr.strip():
        print(proc.stderr)
    os.replace(tmp, LIB_PATH)  # a process that is loading the old file keeps its mapping
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
