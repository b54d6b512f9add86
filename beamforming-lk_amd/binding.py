"""ctypes binding of libawpu_hip.so (include/awpu_hip.h).

Plumbing only: it loads the in-tree HIP library and forwards to the C ABI.  There is no
Python or CPU implementation of the sweep behind it -- if the library cannot be built or
loaded, importing a compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _build

N_SAMPLES = 256
HIST = 1024
ELEMENTS = 64
DATAGRAM_BYTES = 1032  # sizeof(message), src/fpga/receiver.h:24-30

INTERP_LERP, INTERP_FIR8 = 0, 1
MATH_F32_EXACT, MATH_F32_FAST, MATH_BF16_ACC = 0, 1, 2

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_RANGE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6


class Cfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("device", C.c_int32),
        ("n_streams", C.c_int32),
        ("hist", C.c_int32),
        ("n_pixels", C.c_int32),
        ("lut_stride", C.c_int32),
        ("interp", C.c_int32),
        ("math", C.c_int32),
        ("max_batch", C.c_int32),
        ("pixel_begin", C.c_int32),
        ("pixel_count", C.c_int32),
        ("grid_columns", C.c_int32),
        ("n_devices", C.c_int32),
        ("devices", C.c_int32 * 8),
        ("window_begin", C.c_int32),
        ("window_end", C.c_int32),
        ("reserved", C.c_int32 * 1),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("launches", C.c_uint64),
        ("last_kernel_ms", C.c_double),
        ("total_kernel_ms", C.c_double),
        ("alg_bytes_frame", C.c_uint64),
        ("alg_flops_frame", C.c_uint64),
        ("tau_max", C.c_int32),
        ("window", C.c_int32),
        ("usable", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("group_exchange", C.c_int32),
        ("group_ranges", C.c_int32),
    ]


class AwpuError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


PEER_SAME_DEVICE, PEER_DIRECT, PEER_HOST_STAGED = 0, 1, 2
EXCHANGE_NONE, EXCHANGE_WINDOWS, EXCHANGE_PACKED_PAIRS = 0, 1, 2  # Stats.group_exchange
# awpu_kernel_id (include/awpu_hip.h): Stats.kernel_variant after a launch
KERNEL_NAMES = ("none", "quad", "pair", "pair_stationary", "quadh", "quadh_stationary", "single_db", "single_small", "fir8_planes",
                "fir8", "exact_pair", "exact_verify", "tuning", "exact_quad", "exact_nd", "exact_ndh", "exact_ndh_stationary", "exact_ndp")

_lib: Optional[C.CDLL] = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

_SIGNATURES = {
    "awpu_hip_default_cfg": (None, [C.POINTER(Cfg)]),
    "awpu_hip_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Cfg)]),
    "awpu_hip_destroy": (C.c_int, [C.c_void_p]),
    "awpu_hip_set_delay_table": (C.c_int, [C.c_void_p, _i32p, _f32p]),
    "awpu_hip_set_active_mics": (C.c_int, [C.c_void_p, _i32p, C.c_int32]),
    "awpu_hip_set_fir_table": (C.c_int, [C.c_void_p, _f32p]),
    "awpu_hip_process": (C.c_int, [C.c_void_p, _f32p, C.c_int32, _f32p]),
    "awpu_hip_process_async": (C.c_int, [C.c_void_p, _f32p, C.c_int32, _f32p]),
    "awpu_hip_wait": (C.c_int, [C.c_void_p]),
    "awpu_hip_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "awpu_hip_process_device_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "awpu_hip_synchronize": (C.c_int, [C.c_void_p]),
    "awpu_hip_packed_bytes": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "awpu_hip_pack_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "awpu_hip_process_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "awpu_hip_ingest_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "awpu_hip_process_ring": (C.c_int, [C.c_void_p, _f32p]),
    "awpu_hip_ring_snapshot": (C.c_int, [C.c_void_p, _f32p]),
    "awpu_hip_heatmap_u8": (C.c_int, [_f32p, C.c_int32, _u8p]),
    "awpu_hip_live_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, _f32p, C.c_int32, C.c_int32, _u8p, C.c_int32,
                                      C.c_int32, C.c_void_p, _u8p]),
    "awpu_hip_steer_table": (C.c_int, [_f32p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
                                       _i32p, _f32p]),
    "awpu_hip_beams": (C.c_int, [C.c_void_p, C.c_void_p, _i32p, _f32p, C.c_int32, _f32p, _f32p]),
    "awpu_hip_set_mic_gains": (C.c_int, [C.c_void_p, _f32p]),
    "awpu_hip_calibrate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, _i32p, _f32p,
                                            C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p]),
    "awpu_hip_calibrate_ring": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, _i32p, _f32p, C.POINTER(C.c_float),
                                          C.POINTER(C.c_int32)]),
    "awpu_hip_calibrate_host": (C.c_int, [C.c_void_p, _f32p, C.c_int32, C.c_float, _i32p, _f32p, C.POINTER(C.c_float),
                                          C.POINTER(C.c_int32)]),
    "awpu_hip_last_error_of": (C.c_char_p, [C.c_void_p]),
    "awpu_hip_upscale_u8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "awpu_hip_resize_linear_u8": (C.c_int, [_u8p, C.c_int32, C.c_int32, _u8p, C.c_int32, C.c_int32]),
    "awpu_hip_heatmap_u8_device": (
        C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "awpu_hip_create_antenna": (C.c_int, [C.c_int32, C.c_int32, C.c_float, _f32p]),
    "awpu_hip_create_tiled_antenna": (C.c_int, [C.c_int32, C.c_int32, C.c_float, _f32p]),
    "awpu_hip_steering_delays": (C.c_int, [_f32p, C.c_int32, C.c_double, C.c_double, _f32p]),
    "awpu_hip_build_delay_table": (
        C.c_int,
        [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, _i32p, _f32p],
    ),
    "awpu_hip_build_delay_table_device": (
        C.c_int,
        [C.c_int32, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, _i32p, _f32p],
    ),
    "awpu_hip_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "awpu_hip_group_peer_status": (C.c_int, [C.c_void_p, _i32p, C.c_int32]),
    "awpu_hip_strerror": (C.c_char_p, [C.c_int]),
    "awpu_hip_last_error": (C.c_char_p, []),
    "awpu_hip_abi_version": (C.c_int, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load(build: bool = True) -> C.CDLL:
    """Load (building first if stale) the in-tree libawpu_hip.so."""
    global _lib
    if _lib is not None:
        return _lib
    # AWPU_NO_BUILD=1: never start a compiler from this process (set by the profiling scripts: under rocprofv3
    # a child process inherits the profiler's preloaded library, see tools/pmc.sh)
    if os.environ.get("AWPU_NO_BUILD") == "1":
        build = False
    path = _build.build_library() if build else _build.LIB_PATH
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; a process that uses both torch and
    # this library must run on ONE HIP runtime, and the first one loaded wins the SONAME.  Load
    # torch's first when torch is installed, so that libawpu_hip.so binds to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not path.exists():
        raise RuntimeError(f"{path} is missing and there is no CPU fallback")
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(status: int, where: str) -> None:
    if status != OK:
        lib = load()
        detail = lib.awpu_hip_strerror(status).decode()
        last = lib.awpu_hip_last_error().decode()
        raise AwpuError(status, where, f"{detail}; {last}" if last else detail)


def _f32(a: np.ndarray):
    # (the buffer protocol is 5 x cheaper than ndarray.ctypes -- 0.7 against 3.5 us per argument on the build host -- and the one-frame
    # call is ~45 us in all; read-only or empty arrays take the general path)
    try:
        return C.byref(C.c_float.from_buffer(a))
    except (TypeError, ValueError):
        return a.ctypes.data_as(_f32p)


def _i32(a: np.ndarray):
    return a.ctypes.data_as(_i32p)


# ----------------------------------------------------------------------------- geometry


def create_antenna(columns: int = 8, rows: int = 8, distance: float = 0.02) -> np.ndarray:
    """create_antenna, src/geometry/antenna.cpp:60-87 -> xyz[3, rows*columns]."""
    xyz = np.empty((3, rows * columns), np.float32)
    _check(load().awpu_hip_create_antenna(columns, rows, distance, _f32(xyz)), "create_antenna")
    return xyz


def create_tiled_antenna(arrays_x: int, arrays_y: int, distance: float = 0.02) -> np.ndarray:
    xyz = np.empty((3, 64 * arrays_x * arrays_y), np.float32)
    _check(load().awpu_hip_create_tiled_antenna(arrays_x, arrays_y, distance, _f32(xyz)),
           "create_tiled_antenna")
    return xyz


def steering_delays(xyz: np.ndarray, theta: float, phi: float) -> np.ndarray:
    """steering_vector_spherical, src/geometry/antenna.cpp:126-129."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    tau = np.empty(xyz.shape[1], np.float32)
    _check(load().awpu_hip_steering_delays(_f32(xyz), xyz.shape[1], theta, phi, _f32(tau)),
           "steering_delays")
    return tau


def build_delay_table(xyz: np.ndarray, rows: int, columns: int, fov_deg: float = 180.0,
                      row_begin: int = 0, row_count: Optional[int] = None):
    """MIMOWorker::computeDelayLUT, src/dsp/mimo.cpp:20-59 -> (off, frac) [row_count*columns, n]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[1]
    row_count = rows - row_begin if row_count is None else row_count
    off = np.empty((row_count * columns, n), np.int32)
    frac = np.empty((row_count * columns, n), np.float32)
    _check(load().awpu_hip_build_delay_table(_f32(xyz), n, rows, columns, fov_deg, row_begin,
                                             row_count, _i32(off), _f32(frac)), "build_delay_table")
    return off, frac


def build_delay_table_device(xyz: np.ndarray, rows: int, columns: int, fov_deg: float = 180.0,
                             row_begin: int = 0, row_count: Optional[int] = None, device: int = 0):
    """The same table, its rows x columns x n part computed on HIP device `device` (bit-identical to build_delay_table)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[1]
    row_count = rows - row_begin if row_count is None else row_count
    off = np.empty((row_count * columns, n), np.int32)
    frac = np.empty((row_count * columns, n), np.float32)
    _check(load().awpu_hip_build_delay_table_device(device, _f32(xyz), n, rows, columns, fov_deg, row_begin,
                                                    row_count, _i32(off), _f32(frac)), "build_delay_table_device")
    return off, frac


def steer_table(xyz: np.ndarray, theta, phi):
    """Particle::steer (src/dsp/particle.cpp:37-49) for a batch of directions -> (off, frac) [n_dir, n]."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    theta = np.ascontiguousarray(np.atleast_1d(theta), np.float64)
    phi = np.ascontiguousarray(np.atleast_1d(phi), np.float64)
    if theta.shape != phi.shape or theta.ndim != 1:
        raise ValueError("theta and phi must be 1-D and alike")
    n = xyz.shape[1]
    off = np.empty((theta.size, n), np.int32)
    frac = np.empty((theta.size, n), np.float32)
    dp = C.POINTER(C.c_double)
    _check(load().awpu_hip_steer_table(_f32(xyz), n, theta.ctypes.data_as(dp), phi.ctypes.data_as(dp), theta.size,
                                       _i32(off), _f32(frac)), "steer_table")
    return off, frac


def resize_linear_u8(pix: np.ndarray, out_rows: int, out_cols: int) -> np.ndarray:
    """cv::resize(..., INTER_LINEAR) of AWProcessingUnit::draw, src/aw_processing_unit/aw_processing_unit.cpp:252
    (8-bit single channel, upscaling only), on the host."""
    pix = np.ascontiguousarray(pix, np.uint8)
    if pix.ndim != 2:
        raise ValueError("pix must be [rows][cols]")
    out = np.empty((out_rows, out_cols), np.uint8)
    _check(load().awpu_hip_resize_linear_u8(pix.ctypes.data_as(_u8p), pix.shape[0], pix.shape[1],
                                            out.ctypes.data_as(_u8p), out_rows, out_cols), "resize_linear_u8")
    return out


def heatmap_u8(power: np.ndarray) -> np.ndarray:
    """MIMOWorker::populateHeatmap (USE_DB 0), src/dsp/mimo.cpp:61-95."""
    power = np.ascontiguousarray(power, np.float32)
    pix = np.empty(power.shape, np.uint8)
    _check(load().awpu_hip_heatmap_u8(_f32(power), power.size, pix.ctypes.data_as(_u8p)), "heatmap_u8")
    return pix


# ------------------------------------------------------------------------------- engine


class Engine:
    """One awpu_hip handle = one MIMO worker's sweep state on one GPU (src/dsp/mimo.h:74-91)."""

    def __init__(self, n_pixels: int, n_streams: int = ELEMENTS, lut_stride: Optional[int] = None,
                 hist: int = HIST, math: Optional[int] = None, interp: int = INTERP_LERP,
                 max_batch: int = 1, device: int = 0, pixel_begin: int = 0, pixel_count: int = 0,
                 grid_columns: int = 0, devices=None, window=None):
        lib = load()
        cfg = Cfg()
        lib.awpu_hip_default_cfg(C.byref(cfg))
        cfg.device = device
        if devices is not None:  # a device group: the pixels are spread over these GPUs (include/awpu_hip.h)
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices[:8]):  # (more than AWPU_MAX_DEVICES: the library refuses the count)
                cfg.devices[i] = d
            cfg.device = devices[0]
        cfg.grid_columns = grid_columns
        cfg.n_streams = n_streams
        cfg.hist = hist
        cfg.n_pixels = n_pixels
        cfg.lut_stride = n_streams if lut_stride is None else lut_stride
        cfg.interp = interp
        if math is not None:  # None: the library's default (awpu_hip_default_cfg: AWPU_MATH_F32_EXACT, the reference's arithmetic)
            cfg.math = math
        cfg.max_batch = max_batch
        cfg.pixel_begin = pixel_begin
        cfg.pixel_count = pixel_count
        if window is not None:  # (begin, end): history samples staged per stream, the union over the ranks' slabs
            cfg.window_begin, cfg.window_end = int(window[0]), int(window[1])
        self._h = C.c_void_p()
        _check(lib.awpu_hip_create(C.byref(self._h), C.byref(cfg)), "awpu_hip_create")
        self.cfg = cfg
        self.pixel_count = pixel_count or n_pixels
        self._lib = lib

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.awpu_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_delay_table(self, off: np.ndarray, frac: np.ndarray) -> None:
        off = np.ascontiguousarray(off, np.int32)
        frac = np.ascontiguousarray(frac, np.float32)
        want = (self.pixel_count, self.cfg.lut_stride)
        if off.shape != want or frac.shape != want:
            raise ValueError(f"delay tables must be {want}, got {off.shape} / {frac.shape}")
        _check(self._lib.awpu_hip_set_delay_table(self._h, _i32(off), _f32(frac)), "set_delay_table")

    def set_active_mics(self, index: Optional[np.ndarray] = None, usable: Optional[int] = None) -> None:
        if index is None:
            n = self.cfg.n_streams if usable is None else usable
            _check(self._lib.awpu_hip_set_active_mics(self._h, None, n), "set_active_mics")
        else:
            index = np.ascontiguousarray(index, np.int32)
            _check(self._lib.awpu_hip_set_active_mics(self._h, _i32(index), index.size),
                   "set_active_mics")

    def ingest_block(self, datagrams) -> None:
        """One block of 256 wire datagrams (bytes-like, 256 x 1032 B; src/fpga/receiver.h:24-30)."""
        buf = np.frombuffer(datagrams, dtype=np.uint8)
        if buf.size != 256 * DATAGRAM_BYTES:
            raise ValueError("a block is 256 datagrams of 1032 bytes")
        _check(self._lib.awpu_hip_ingest_block(self._h, buf.ctypes.data_as(C.c_void_p), DATAGRAM_BYTES), "ingest_block")

    def process_ring(self) -> np.ndarray:
        power = np.empty(self.pixel_count, np.float32)
        _check(self._lib.awpu_hip_process_ring(self._h, _f32(power)), "process_ring")
        return power

    def ring_snapshot(self) -> np.ndarray:
        frames = np.empty((self.cfg.n_streams, HIST), np.float32)
        _check(self._lib.awpu_hip_ring_snapshot(self._h, _f32(frames)), "ring_snapshot")
        return frames

    def set_fir_table(self, coeffs: np.ndarray) -> None:
        """The caller's [101, 8] coefficient table of the FIR variant (src/dsp/filter.h:10-112)."""
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        if coeffs.shape != (101, 8):
            raise ValueError("FIR table must be [101, 8]")
        _check(self._lib.awpu_hip_set_fir_table(self._h, _f32(coeffs)), "set_fir_table")

    def process(self, frames: np.ndarray) -> np.ndarray:
        """frames [batch, n_streams, hist] (or one frame [n_streams, hist]) -> power [batch, pixels]."""
        frames = np.ascontiguousarray(frames, np.float32)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        if frames.shape[1:] != (self.cfg.n_streams, self.cfg.hist):
            raise ValueError(f"frames must be [batch, {self.cfg.n_streams}, {self.cfg.hist}]")
        power = np.empty((frames.shape[0], self.pixel_count), np.float32)
        _check(self._lib.awpu_hip_process(self._h, _f32(frames), frames.shape[0], _f32(power)),
               "awpu_hip_process")
        return power[0] if single else power

    def process_async(self, frames: np.ndarray):
        """awpu_hip_process_async: enqueue upload + sweep + read-back and return; wait() hands out the power.
        The arrays are kept alive (and must not be touched) until wait() returns."""
        frames = np.ascontiguousarray(frames, np.float32)
        if frames.ndim != 3 or frames.shape[1:] != (self.cfg.n_streams, self.cfg.hist):
            raise ValueError(f"frames must be [batch, {self.cfg.n_streams}, {self.cfg.hist}]")
        power = np.empty((frames.shape[0], self.pixel_count), np.float32)
        _check(self._lib.awpu_hip_process_async(self._h, _f32(frames), frames.shape[0], _f32(power)), "awpu_hip_process_async")
        self._pending = (frames, power)

    def wait(self) -> Optional[np.ndarray]:
        _check(self._lib.awpu_hip_wait(self._h), "awpu_hip_wait")
        pending, self._pending = getattr(self, "_pending", None), None
        return pending[1] if pending else None

    def process_device(self, d_frames_ptr: int, batch: int, d_power_ptr: int, stream: int = 0) -> None:
        """Asynchronous sweep on device pointers (e.g. torch tensors' data_ptr()) on `stream`."""
        _check(self._lib.awpu_hip_process_device(self._h, C.c_void_p(d_frames_ptr), batch,
                                                 C.c_void_p(d_power_ptr), C.c_void_p(stream)),
               "awpu_hip_process_device")

    def process_device_sums(self, d_frames_ptr: int, batch: int, d_power_ptr: int, d_sums_ptr: int, stream: int = 0) -> None:
        """process_device plus every pixel's out[0..255] before the epilogue into d_sums [batch, pixels, 256]
        (awpu_hip_process_device_sums: MATH_F32_EXACT + INTERP_LERP only; bit-identical to the reference's out[])."""
        _check(self._lib.awpu_hip_process_device_sums(self._h, C.c_void_p(d_frames_ptr), batch, C.c_void_p(d_power_ptr),
                                                      C.c_void_p(d_sums_ptr), C.c_void_p(stream)),
               "awpu_hip_process_device_sums")

    def packed_bytes(self, batch: int) -> int:
        """Bytes of the packed frame-pair buffer for `batch` frames (awpu_hip_packed_bytes); raises AwpuError with
        status ERR_STATE when this handle's sweep does not take packed frames."""
        n = C.c_uint64(0)
        _check(self._lib.awpu_hip_packed_bytes(self._h, batch, C.byref(n)), "awpu_hip_packed_bytes")
        return int(n.value)

    def pack_frames(self, d_frames_ptr: int, batch: int, d_packed_ptr: int, stream: int = 0) -> None:
        """The sweep's pack pass on its own (the ingest rank of a multi-GPU job): frames -> packed pairs, on `stream`."""
        _check(self._lib.awpu_hip_pack_frames(self._h, C.c_void_p(d_frames_ptr), batch, C.c_void_p(d_packed_ptr),
                                              C.c_void_p(stream)), "awpu_hip_pack_frames")

    def process_packed(self, d_packed_ptr: int, batch: int, d_power_ptr: int, stream: int = 0) -> None:
        """The sweep on packed frame pairs as they arrive from the ingest rank, on `stream`; asynchronous."""
        _check(self._lib.awpu_hip_process_packed(self._h, C.c_void_p(d_packed_ptr), batch, C.c_void_p(d_power_ptr),
                                                 C.c_void_p(stream)), "awpu_hip_process_packed")

    def heatmap_device(self, d_power_ptr: int, n: int, batch: int, d_peak_ptr: int, d_pix_ptr: int,
                       peak_given: bool = False, stream: int = 0) -> None:
        """populateHeatmap (src/dsp/mimo.cpp:61-95) on device buffers, asynchronous on `stream`."""
        _check(self._lib.awpu_hip_heatmap_u8_device(self._h, C.c_void_p(d_power_ptr), n, batch,
                                                    C.c_void_p(d_peak_ptr), int(peak_given), C.c_void_p(d_pix_ptr),
                                                    C.c_void_p(stream)), "awpu_hip_heatmap_u8_device")

    def beams(self, off: np.ndarray, frac: np.ndarray, d_frame_ptr: int = 0, want_beams: bool = True):
        """Particle::beam / Particle::das (src/dsp/particle.cpp:51-103) for off/frac [n_dir, lut_stride];
        d_frame_ptr 0 = the ingest ring's snapshot -> (power [n_dir], beams [n_dir, 256] or None)."""
        off = np.ascontiguousarray(off, np.int32)
        frac = np.ascontiguousarray(frac, np.float32)
        if off.ndim != 2 or off.shape[1] != self.cfg.lut_stride or frac.shape != off.shape:
            raise ValueError("off/frac must be [n_dir, lut_stride]")
        n = off.shape[0]
        power = np.empty(n, np.float32)
        out = np.empty((n, 256), np.float32) if want_beams else None
        _check(self._lib.awpu_hip_beams(self._h, C.c_void_p(d_frame_ptr), _i32(off), _f32(frac), n, _f32(power),
                                        _f32(out) if want_beams else None), "awpu_hip_beams")
        return power, out

    def live_block(self, wire, rows: int, cols: int, out_rows: int = 0, out_cols: int = 0,
                   d_colormap_ptr: int = 0, want_power: bool = True, out=None):
        """Block in, images out (awpu_hip_live_block): ingest 256 raw datagrams, sweep the new snapshot, 8-bit
        heatmap, optional upscale -> (power or None, image [rows, cols], big image or None).  `wire`: bytes or a
        contiguous uint8 array (a receive buffer that is refilled in place); `out` = (power, image, big) arrays of an
        earlier call to write into again (a display loop keeps its buffers: the library then replays one HIP graph
        per ring position instead of enqueuing the steps one by one)."""
        if len(wire) != 256 * 1032:
            raise ValueError("wire must be 256 datagrams of 1032 bytes")
        if isinstance(wire, np.ndarray):
            if wire.dtype != np.uint8 or not wire.flags.c_contiguous:
                raise ValueError("wire array must be contiguous uint8")
            wire = wire.ctypes.data_as(C.c_void_p)
        if out is not None:
            power, image, big = out
        else:
            power = np.empty(self.cfg.n_pixels, np.float32) if want_power else None
            image = np.empty((rows, cols), np.uint8)
            big = None
            if out_rows:
                big = np.empty((out_rows, out_cols, 3) if d_colormap_ptr else (out_rows, out_cols), np.uint8)
        want_power = power is not None
        _check(self._lib.awpu_hip_live_block(self._h, wire, 1032, _f32(power) if want_power else None, rows, cols,
                                             image.ctypes.data_as(_u8p), out_rows, out_cols, C.c_void_p(d_colormap_ptr),
                                             big.ctypes.data_as(_u8p) if big is not None else None), "awpu_hip_live_block")
        return power, image, big

    def set_mic_gains(self, gains: Optional[np.ndarray]) -> None:
        """Optional per-mic gain (the reference's unused power_correction_mask, aw_processing_unit.cpp:190-200);
        gains [n_streams] by stream id, None = off."""
        if gains is None:
            _check(self._lib.awpu_hip_set_mic_gains(self._h, None), "set_mic_gains")
            return
        gains = np.ascontiguousarray(gains, np.float32)
        if gains.shape != (self.cfg.n_streams,):
            raise ValueError("gains must be [n_streams]")
        _check(self._lib.awpu_hip_set_mic_gains(self._h, _f32(gains)), "set_mic_gains")

    def _calibrated(self, call):
        index = np.empty(64, np.int32)
        corr = np.empty(64, np.float32)
        med, n = C.c_float(0), C.c_int32(0)
        call(index, corr, med, n)
        return index[: n.value].copy(), corr[: n.value].copy(), med.value

    def calibrate_device(self, d_frame_ptr: int, array: int = 0, reference_power_level: float = 1e-5, stream: int = 0):
        """AWProcessingUnit::calibrate (aw_processing_unit.cpp:102-212) for one array of a snapshot in
        device memory -> (index, correction, median)."""
        return self._calibrated(lambda i, c, m, n: _check(self._lib.awpu_hip_calibrate_device(
            self._h, C.c_void_p(d_frame_ptr), array, reference_power_level, _i32(i), _f32(c), C.byref(m), C.byref(n),
            C.c_void_p(stream)), "awpu_hip_calibrate_device"))

    def calibrate_host(self, frame: np.ndarray, array: int = 0, reference_power_level: float = 1e-5):
        """The same for a snapshot [n_streams, hist] in host memory."""
        frame = np.ascontiguousarray(frame, np.float32)
        if frame.shape != (self.cfg.n_streams, self.cfg.hist):
            raise ValueError("frame must be [n_streams, hist]")
        return self._calibrated(lambda i, c, m, n: _check(self._lib.awpu_hip_calibrate_host(
            self._h, _f32(frame), array, reference_power_level, _i32(i), _f32(c), C.byref(m), C.byref(n)),
            "awpu_hip_calibrate_host"))

    def last_error(self) -> str:
        return self._lib.awpu_hip_last_error_of(self._h).decode()

    def calibrate_ring(self, array: int = 0, reference_power_level: float = 1e-5):
        """The same on the current snapshot of the ingest ring."""
        return self._calibrated(lambda i, c, m, n: _check(self._lib.awpu_hip_calibrate_ring(
            self._h, array, reference_power_level, _i32(i), _f32(c), C.byref(m), C.byref(n)), "awpu_hip_calibrate_ring"))

    def upscale_device(self, d_pix_ptr: int, rows: int, cols: int, batch: int, d_out_ptr: int, out_rows: int,
                       out_cols: int, d_colormap_ptr: int = 0, stream: int = 0) -> None:
        """The display upscale (aw_processing_unit.cpp:252), optionally through a 256x3 colour table
        (main.cpp:345), on device buffers; asynchronous on `stream`."""
        _check(self._lib.awpu_hip_upscale_u8_device(self._h, C.c_void_p(d_pix_ptr), rows, cols, batch,
                                                    C.c_void_p(d_colormap_ptr), C.c_void_p(d_out_ptr), out_rows,
                                                    out_cols, C.c_void_p(stream)), "awpu_hip_upscale_u8_device")

    def synchronize(self) -> None:
        _check(self._lib.awpu_hip_synchronize(self._h), "awpu_hip_synchronize")

    def peer_status(self) -> list:
        """Per device of the group: PEER_SAME_DEVICE / PEER_DIRECT / PEER_HOST_STAGED (awpu_hip_group_peer_status)."""
        buf = (C.c_int32 * 8)()
        n = self._lib.awpu_hip_group_peer_status(self._h, buf, 8)
        if n < 0:
            _check(n, "group_peer_status")
        return [int(buf[k]) for k in range(n)]

    def stats(self) -> Stats:
        st = Stats()
        _check(self._lib.awpu_hip_get_stats(self._h, C.byref(st)), "get_stats")
        return st
