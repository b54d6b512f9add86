"""Randomised GPU-vs-oracle parity with the kernel shape forced each way (one child process per
shape, because the library reads AWPU_SHAPE once per process)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [
    {"AWPU_SHAPE": "pair"},                                     # frame-pair sweep on every batch >= 2
    {"AWPU_SHAPE": "single_db"},                                # double-buffered single-frame shape
    {"AWPU_SHAPE": "single_small"},                             # 8-wave shape, 2 pixels per wave
    {"AWPU_TEST_MATH": "exact"},                                # the reference-order kernel on the frame-pair layout
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},  # the default dispatch on grids: das_exact_nd_kernel for batches, das_exact_ndh_kernel (resident / 4-, 8-, 16-wave chunked) for single frames
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1", "AWPU_SHAPE": "exact_pair"},  # ... two-pixel block on the same tables
    {"AWPU_TEST_MATH": "exact", "AWPU_SHAPE": "exact_verify"},  # the round-1 verification kernel (the bf16 mode's structure)
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_SHAPE": "exact_nd2"},  # the {next, d} kernel on RANDOM delays: every pixel leaves the reference's address (its read-on-the-spot paths), batches of one as a pair with itself
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1", "AWPU_SHAPE": "exact_nd1"},  # ... one quad per wave, delays that mostly coincide
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_TEST_BATCH": "1", "AWPU_SHAPE": "exact_ndp"},  # single frames: one pixel per wave, on random delays, every case one frame per call
    {"AWPU_TEST_MATH": "exact", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1", "AWPU_SHAPE": "exact_quad"},  # round 4's quad kernel on raw sample pairs (still the fallback)
    {"AWPU_TEST_PATH": "device"},                               # device-resident frames + two pixel shards per case
    {"AWPU_TEST_INTERP": "fir8"},                               # the 8-tap variant of delay()
    {"AWPU_TEST_INTERP": "fir8", "AWPU_SHAPE": "fir8_planes"},  # ... on the four-plane frame-pair kernel for every batch >= 2
    {"AWPU_TEST_REUSE": "1"},                                   # one handle re-targeted: tables, mic lists, gains
    {"AWPU_SHAPE": "pair_vertical", "AWPU_TEST_GRID": "1"},     # frame pairs, vertical pixel pairs
    {"AWPU_SHAPE": "quad", "AWPU_TEST_GRID": "1"},                               # quad shapes, random delays: every pixel differs
    {"AWPU_SHAPE": "quad", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},    # quad shapes, delays that mostly coincide
    {"AWPU_SHAPE": "pair", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},    # pair shape on the same tables
    {"AWPU_SHAPE": "stationary"},                                                # stationary pair shape wherever the window fits the LDS
    {"AWPU_SHAPE": "stationary", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},
    {"AWPU_SHAPE": "quadh", "AWPU_TEST_GRID": "1"},                              # single-frame quad shape on the halves layout for every call
    {"AWPU_SHAPE": "quadh", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},
    {"AWPU_SHAPE": "quadh_chunked", "AWPU_TEST_GRID": "1", "AWPU_TEST_COINCIDE": "1"},  # ... never its resident-window variant
], ids=["pairs", "db", "small", "exact", "exact_grid", "exact_grid_pairs", "exact_verify", "exact_nd2_random", "exact_nd1_coincide", "exact_ndp_random", "exact_quad_r4", "device", "fir8", "fir8_planes", "reuse", "pairs_vertical",
        "quads_random", "quads_coincide", "pairs_coincide", "stationary", "stationary_grid", "quadh_every_call", "quadh_coincide",
        "quadh_chunked"])
def test_random_tables(env):
    out = subprocess.run([sys.executable, str(REPO / "tests" / "gpu_random_check.py"), "2024", "14"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK 14 cases" in out.stdout, out.stdout + out.stderr[-2000:]
