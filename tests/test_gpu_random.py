"""Randomised GPU-vs-oracle parity with the kernel shape forced each way (one child process per
shape, because the library reads AWPU_FAST_* once per process)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [
    {"AWPU_FAST_PAIRS": "1"},                                  # frame-pair sweep on every batch >= 2
    {"AWPU_FAST_PAIRS": "0", "AWPU_FAST_VARIANT": "1,8,32"},   # double-buffered single-frame shape
    {"AWPU_FAST_PAIRS": "0", "AWPU_FAST_VARIANT": "1,2,8"},    # 8-wave shape, 2 pixels per wave
    {"AWPU_FAST_PAIRS": "0", "AWPU_FAST_VARIANT": "2,4,8"},    # two frames per item, compiler-scheduled
    {"AWPU_TEST_MATH": "exact"},                                # the exact-order kernel
    {"AWPU_TEST_PATH": "device"},                               # device-resident frames + two pixel shards per case
    {"AWPU_TEST_INTERP": "fir8"},                               # the 8-tap variant of delay()
    {"AWPU_TEST_INTERP": "fir8", "AWPU_FIR8_PLANES": "2"},      # ... on the four-plane frame-pair kernel for every batch >= 2
    {"AWPU_TEST_REUSE": "1"},                                   # one handle re-targeted: tables, mic lists, gains
    {"AWPU_FAST_PAIRS": "1", "AWPU_TEST_GRID": "1", "AWPU_FAST_PAIRCOLS": "1"},  # frame pairs, vertical pixel pairs
    {"AWPU_FAST_PAIRS": "1", "AWPU_TEST_GRID": "1", "AWPU_FAST_DEBUG": "4096"},   # the block without read sharing
    {"AWPU_FAST_QUADS": "1", "AWPU_TEST_GRID": "1"},                               # quad shape, random delays: every pixel differs
    {"AWPU_FAST_QUADS": "1", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},    # quad shape, delays that mostly coincide
    {"AWPU_FAST_QUADS": "0", "AWPU_FAST_PAIRS": "1", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},  # pair shape on the same tables
    {"AWPU_FAST_STATIONARY": "1", "AWPU_FAST_PAIRS": "1"},                          # stationary pair shape wherever the window fits the LDS
    {"AWPU_FAST_STATIONARY": "1", "AWPU_FAST_PAIRS": "1", "AWPU_TEST_GRID": "1", "AWPU_FAST_QUADS": "0", "AWPU_TEST_COINCIDE": "1"},
    {"AWPU_FAST_QUADS": "1", "AWPU_TEST_GRID": "1", "AWPU_FAST_PAIRS": "0"},                             # single-frame quad shape on the halves layout for every call
    {"AWPU_FAST_QUADS": "1", "AWPU_TEST_GRID": "1", "AWPU_FAST_PAIRS": "0", "AWPU_FAST_HALVES": "0", "AWPU_TEST_COINCIDE": "1"},  # ... round 2's in-place kernel
], ids=["pairs", "db", "small", "fpi2", "exact", "device", "fir8", "fir8_planes", "reuse", "pairs_vertical", "pairs_unshared",
        "quads_random", "quads_coincide", "pairs_coincide", "stationary", "stationary_grid", "quadh_every_call", "quad1_in_place"])
def test_random_tables(env):
    out = subprocess.run([sys.executable, str(REPO / "tests" / "gpu_random_check.py"), "2024", "14"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK 14 cases" in out.stdout, out.stdout + out.stderr[-2000:]
