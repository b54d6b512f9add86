"""Seeded synthetic workloads for the delay-and-sum heatmap path.

A frame is the snapshot MIMOWorker::update takes (src/dsp/mimo.cpp:100-103): for every
mic stream 1024 floats, oldest..newest (src/fpga/streams.hpp:113-116), values in (-1, 1)
like the normalised 24-bit samples of src/fpga/pipeline.cpp:290.  The content follows the
reference's synthetic producer (src/fpga/pipeline.cpp:105-135): a 9 kHz plane wave of
amplitude 1e-2, here from an off-axis direction, plus uniform noise.

Geometry for more than one 8x8 array is build-defined (the reference only beamforms
antennas[0]): arrays tiled at the same pitch, see DESIGN.md.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import binding

SAMPLE_RATE = 48828.0  # src/geometry/antenna.h:17
CARRIER = 9e3          # src/fpga/pipeline.cpp:117
AMPLITUDE = 1e-2       # src/fpga/pipeline.cpp:133
NOISE = 1e-3
SOURCE_THETA = np.deg2rad(20.0)
SOURCE_PHI = np.deg2rad(35.0)


@dataclass(frozen=True)
class WorkloadSpec:
    name: str
    arrays_x: int
    arrays_y: int
    res: int  # the heatmap is res x res (AWProcessingUnit small_res, aw_processing_unit.cpp:74)
    fov: float = 180.0

    @property
    def n_mics(self) -> int:
        return 64 * self.arrays_x * self.arrays_y

    @property
    def n_pixels(self) -> int:
        return self.res * self.res


# BASELINE.json configs; "headline" is the shape the north-star target is quoted on.
WORKLOADS = {
    "c1": WorkloadSpec("c1: 64 mics x 32x32 x 256", 1, 1, 32),
    "c2": WorkloadSpec("c2: 256 mics x 64x64 x 256", 4, 1, 64),
    "headline": WorkloadSpec("headline: 256 mics x 128x128 x 256", 4, 1, 128),
    "c3": WorkloadSpec("c3: 512 mics x 128x128 x 256", 4, 2, 128),
    "c4": WorkloadSpec("c4: 512 mics x 256x256 x 256", 4, 2, 256),
    # what the reference ships: one 8x8 array, --mimo-res 100, --fov 180 (src/main.cpp:38-41,53-56)
    "ref_default": WorkloadSpec("reference default: 64 mics x 100x100 x 256", 1, 1, 100),
    # the FPGA's four arrays at the reference's default resolution (not a BASELINE config: the mid-size single-frame case)
    "ref_4arrays": WorkloadSpec("four arrays at the reference's resolution: 256 mics x 100x100 x 256", 4, 1, 100),
}


def geometry(spec: WorkloadSpec) -> np.ndarray:
    return binding.create_tiled_antenna(spec.arrays_x, spec.arrays_y)


def delay_table(spec: WorkloadSpec, xyz: Optional[np.ndarray] = None, row_begin: int = 0,
                row_count: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    xyz = geometry(spec) if xyz is None else xyz
    return binding.build_delay_table(xyz, spec.res, spec.res, spec.fov, row_begin, row_count)


def delay_table_for(spec: WorkloadSpec, xyz: Optional[np.ndarray], row_ranges) -> Tuple[np.ndarray, np.ndarray]:
    """The table rows of a shard (sharding.RowShard.row_ranges: contiguous or interleaved), concatenated in the
    order the shard's power tile holds them."""
    parts = [delay_table(spec, xyz, b, n) for b, n in row_ranges]
    if len(parts) == 1:
        return parts[0]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def make_frames(xyz: np.ndarray, batch: int, seed: int = 1234, theta: float = SOURCE_THETA,
                phi: float = SOURCE_PHI, hist: int = binding.HIST) -> np.ndarray:
    """[batch, n_mics, hist] float32: plane wave from (theta, phi) + uniform noise.

    Mic m hears the wave advanced by its steering delay tau_m (samples), so steering the
    array to (theta, phi) -- which reads each mic tau_m samples earlier -- adds coherently.
    Consecutive frames continue the same wave, one 256-sample block later each.
    """
    n = xyz.shape[1]
    tau = binding.steering_delays(xyz, theta, phi).astype(np.float64)
    # the noise: the raw 32-bit stream of std::mt19937(seed) -- numpy's legacy MT19937 seeding is init_genrand,
    # the same generator (tests/test_host_mirror.py compares it with a g++-compiled std::mt19937) -- drawn frame by frame, mic by
    # mic, sample by sample, each draw r mapped to NOISE * (2 r / 2^32 - 1): reproducible from C++ without libstdc++'s
    # distribution classes
    rng = np.random.RandomState(seed)
    t = np.arange(hist, dtype=np.float64)
    out = np.empty((batch, n, hist), np.float32)
    for b in range(batch):
        phase = 2.0 * np.pi * CARRIER * (t[None, :] + b * binding.N_SAMPLES + tau[:, None]) / SAMPLE_RATE
        raw = rng.randint(0, 2 ** 32, size=(n, hist), dtype=np.uint32).astype(np.float64)
        noise = NOISE * (raw * (2.0 / 4294967296.0) - 1.0)
        out[b] = (AMPLITUDE * np.sin(phase) + noise).astype(np.float32)
    return out


def source_pixel(spec: WorkloadSpec, theta: float = SOURCE_THETA, phi: float = SOURCE_PHI) -> Tuple[int, int]:
    """(row, column) of the grid cell whose direction is closest to (theta, phi).

    Inverse of the sine-space grid of MIMOWorker::computeDelayLUT (src/dsp/mimo.cpp:23-43):
    x_c = (c - res/2 + 0.5) * sep = sin(theta) cos(phi), y_r likewise with sin(phi).
    """
    sep = np.sin(np.deg2rad(spec.fov) / 2.0) / (spec.res / 2.0)
    x = np.sin(theta) * np.cos(phi)
    y = np.sin(theta) * np.sin(phi)
    c = int(round(x / sep + spec.res / 2.0 - 0.5))
    r = int(round(y / sep + spec.res / 2.0 - 0.5))
    return r, c


def algorithmic_bytes_per_frame(n_mics: int, n_pixels: int, window: int) -> int:
    """B_alg = 4 M W + 8 P M + 4 P: staged window, reference-format table, power out."""
    return 4 * n_mics * window + 8 * n_pixels * n_mics + 4 * n_pixels


def algorithmic_flops_per_frame(n_mics: int, n_pixels: int) -> int:
    """F_alg = 4 P M 256 + 6 P 254."""
    return 4 * n_pixels * n_mics * 256 + 6 * n_pixels * 254
