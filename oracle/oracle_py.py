"""ctypes access to the CPU checker: oracle/liboracle_das.so (the C restatement) and
oracle/_ref/libref_das_*.so (the reference's own delay.cpp compiled in place).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under beamforming-lk_amd/ may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_LIB = HERE / "liboracle_das.so"
REF_AVX2_LIB = HERE / "_ref" / "libref_das_avx2.so"
REF_FIR_LIB = HERE / "_ref" / "libref_das_fir.so"
REFERENCE_TREE = Path("/root/reference")

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def build(ref: bool = True) -> None:
    """make the restatement, and (when the reference tree is present) oracle/_ref."""
    if os.environ.get("AWPU_NO_BUILD") == "1":  # set by the profiling scripts: no child processes under rocprofv3
        raise RuntimeError("AWPU_NO_BUILD=1: the oracle library is missing or stale and may not be built from this process")
    subprocess.run(["make", "-s", "-C", str(HERE), "oracle"], check=True)
    if ref and (REFERENCE_TREE / "src/dsp/delay.cpp").exists():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


_oracle: Optional[C.CDLL] = None
_refs: dict = {}


def oracle() -> C.CDLL:
    global _oracle
    if _oracle is None:
        src_m = max((HERE / "das_oracle.c").stat().st_mtime, (HERE / "das_oracle.h").stat().st_mtime)
        if not ORACLE_LIB.exists() or ORACLE_LIB.stat().st_mtime < src_m:
            build(ref=False)
        lib = C.CDLL(str(ORACLE_LIB))
        lib.oracle_create_antenna.argtypes = [C.c_int, C.c_int, C.c_float, _f32p]
        lib.oracle_create_tiled_antenna.argtypes = [C.c_int, C.c_int, C.c_float, _f32p]
        lib.oracle_steering_delays_f32.argtypes = [_f32p, C.c_int, C.c_double, C.c_double, _f32p]
        lib.oracle_steering_delays_f64.argtypes = [_f32p, C.c_int, C.c_double, C.c_double, _f64p]
        lib.oracle_compute_delay_lut.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, _i32p, _f32p]
        lib.oracle_compute_delays_f64.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, _f64p]
        lib.oracle_delay_lerp.argtypes = [_f32p, _f32p, C.c_float]
        lib.oracle_delay_fir8.argtypes = [_f32p, _f32p, C.c_float, _f32p]
        lib.oracle_das_f32.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p, _f32p]
        lib.oracle_das_bf16acc.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p]
        lib.oracle_das_bf16acc.restype = None
        lib.oracle_das_f64.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f64p]
        lib.oracle_das_fir8_f32.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p, _f32p]
        lib.oracle_das_fir8_f64.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p, _f64p]
        lib.oracle_particle_beams.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p, _f32p]
        lib.oracle_particle_beams.restype = None
        lib.oracle_heatmap_u8.argtypes = [_f32p, C.c_int, _u8p]
        lib.oracle_resize_linear_u8.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        lib.oracle_resize_linear_u8.restype = C.c_int
        lib.oracle_calibrate.argtypes = [_f32p, C.c_int, C.c_float, _i32p, _f32p, _f32p]
        lib.oracle_calibrate.restype = C.c_int
        lib.oracle_unpack_exposure.argtypes = [_i32p, C.c_int, C.c_int, _f32p]
        for name in ("oracle_create_antenna", "oracle_create_tiled_antenna", "oracle_steering_delays_f32",
                     "oracle_steering_delays_f64", "oracle_compute_delay_lut", "oracle_compute_delays_f64",
                     "oracle_delay_lerp", "oracle_delay_fir8", "oracle_das_f32", "oracle_das_f64",
                     "oracle_das_fir8_f32", "oracle_das_fir8_f64", "oracle_heatmap_u8", "oracle_unpack_exposure"):
            getattr(lib, name).restype = None
        _oracle = lib
    return _oracle


def ref_available(variant: str = "avx2") -> bool:
    return (REF_AVX2_LIB if variant == "avx2" else REF_FIR_LIB).exists()


def ref(variant: str = "avx2") -> C.CDLL:
    """The reference's compiled delay() + loop-nest driver.  Raises if oracle/_ref is absent."""
    if variant not in _refs:
        path = REF_AVX2_LIB if variant == "avx2" else REF_FIR_LIB
        if not path.exists():
            build(ref=True)
        if not path.exists():
            raise FileNotFoundError(f"{path} missing and {REFERENCE_TREE} not available to build it")
        lib = C.CDLL(str(path))
        lib.ref_variant.restype = C.c_int
        lib.ref_delay.argtypes = [_f32p, _f32p, C.c_float]
        lib.ref_delay.restype = None
        lib.ref_das.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p, _f32p]
        lib.ref_das.restype = None
        lib.ref_particle_beams.argtypes = lib.ref_das.argtypes
        lib.ref_particle_beams.restype = None
        lib.ref_das_bench.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p,
                                      C.c_double, C.POINTER(C.c_int)]
        lib.ref_das_bench.restype = C.c_double
        lib.ref_das_bench_mt.argtypes = [_f32p, C.c_int, _i32p, _f32p, C.c_int, C.c_int, _i32p, C.c_int, _f32p,
                                         C.c_double, C.c_int, C.POINTER(C.c_int)]
        lib.ref_das_bench_mt.restype = C.c_double
        _refs[variant] = lib
    return _refs[variant]


def _p32(a):
    return a.ctypes.data_as(_f32p)


def _pi(a):
    return a.ctypes.data_as(_i32p)


# ------------------------------------------------------------------ numpy-level helpers


def create_antenna(columns=8, rows=8, distance=0.02) -> np.ndarray:
    xyz = np.empty((3, rows * columns), np.float32)
    oracle().oracle_create_antenna(columns, rows, distance, _p32(xyz))
    return xyz


def create_tiled_antenna(arrays_x, arrays_y, distance=0.02) -> np.ndarray:
    xyz = np.empty((3, 64 * arrays_x * arrays_y), np.float32)
    oracle().oracle_create_tiled_antenna(arrays_x, arrays_y, distance, _p32(xyz))
    return xyz


def steering_delays_f32(xyz, theta, phi) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, np.float32)
    tau = np.empty(xyz.shape[1], np.float32)
    oracle().oracle_steering_delays_f32(_p32(xyz), xyz.shape[1], theta, phi, _p32(tau))
    return tau


def steering_delays_f64(xyz, theta, phi) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, np.float32)
    tau = np.empty(xyz.shape[1], np.float64)
    oracle().oracle_steering_delays_f64(_p32(xyz), xyz.shape[1], theta, phi, tau.ctypes.data_as(_f64p))
    return tau


def compute_delay_lut(xyz, rows, columns, fov_deg=180.0):
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[1]
    off = np.empty((rows * columns, n), np.int32)
    frac = np.empty((rows * columns, n), np.float32)
    oracle().oracle_compute_delay_lut(_p32(xyz), n, rows, columns, fov_deg, _pi(off), _p32(frac))
    return off, frac


def compute_delays_f64(xyz, rows, columns, fov_deg=180.0) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[1]
    tau = np.empty((rows * columns, n), np.float64)
    oracle().oracle_compute_delays_f64(_p32(xyz), n, rows, columns, fov_deg, tau.ctypes.data_as(_f64p))
    return tau


def _sweep_args(X, off, frac, index):
    X = np.ascontiguousarray(X, np.float32)
    off = np.ascontiguousarray(off, np.int32)
    frac = np.ascontiguousarray(frac, np.float32)
    if index is None:
        index = np.arange(X.shape[0], dtype=np.int32)
    index = np.ascontiguousarray(index, np.int32)
    assert X.ndim == 2 and off.shape == frac.shape and off.ndim == 2
    assert index.max() < min(X.shape[0], off.shape[1])
    assert off[:, index].min() >= 0 and off[:, index].max() + 257 <= X.shape[1]
    return X, off, frac, index


def das_f32(X, off, frac, index=None, want_out=False, impl="oracle"):
    """power[P] (and out[P,256]) for one frame X[n_streams, hist]; impl = oracle | ref."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    P = off.shape[0]
    power = np.empty(P, np.float32)
    out = np.empty((P, 256), np.float32) if want_out else None
    fn = oracle().oracle_das_f32 if impl == "oracle" else ref("avx2").ref_das
    fn(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index), index.size,
       _p32(power), _p32(out) if want_out else None)
    return (power, out) if want_out else power


def particle_beams(X, off, frac, index=None, impl="oracle"):
    """Particle::beam / Particle::das (particle.cpp:51-103) for off/frac [n_dir, stride] -> (power[n_dir], beams[n_dir, 256])."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    n = off.shape[0]
    power = np.empty(n, np.float32)
    beams = np.empty((n, 256), np.float32)
    fn = oracle().oracle_particle_beams if impl == "oracle" else ref("avx2").ref_particle_beams
    fn(_p32(X), X.shape[1], _pi(off), _p32(frac), n, off.shape[1], _pi(index), index.size, _p32(power), _p32(beams))
    return power, beams


def das_bf16acc(X, off, frac, index=None) -> np.ndarray:
    """The build's bf16-accumulator mode (AWPU_MATH_BF16_ACC), restated: power[P] for one frame."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    P = off.shape[0]
    power = np.empty(P, np.float32)
    oracle().oracle_das_bf16acc(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index), index.size,
                                _p32(power))
    return power


def das_f64(X, off, frac, index=None) -> np.ndarray:
    X, off, frac, index = _sweep_args(X, off, frac, index)
    P = off.shape[0]
    power = np.empty(P, np.float64)
    oracle().oracle_das_f64(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index),
                            index.size, power.ctypes.data_as(_f64p))
    return power


def das_fir8_f32(X, off, frac, coeffs, index=None, impl="oracle"):
    """FIR8 sweep (delay.cpp:31-40 inside mimo.cpp:121-151); impl = oracle | ref (the reference's own
    non-AVX2 build, which carries its own table: `coeffs` must then be that table)."""
    X = np.ascontiguousarray(X, np.float32)
    off = np.ascontiguousarray(off, np.int32)
    frac = np.ascontiguousarray(frac, np.float32)
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    assert coeffs.shape == (101, 8)
    if index is None:
        index = np.arange(X.shape[0], dtype=np.int32)
    index = np.ascontiguousarray(index, np.int32)
    assert off[:, index].min() >= 0 and off[:, index].max() + 263 <= X.shape[1]
    P = off.shape[0]
    power = np.empty(P, np.float32)
    if impl == "oracle":
        oracle().oracle_das_fir8_f32(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index),
                                     index.size, _p32(coeffs), _p32(power))
    else:
        lib = ref("fir")
        assert lib.ref_variant() == 2
        lib.ref_das(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index), index.size,
                    _p32(power), None)
    return power


def das_fir8_f64(X, off, frac, coeffs, index=None) -> np.ndarray:
    """The FIR8 sweep with every sum in double (tie-breaker: how far is the reference's fp32 from exact)."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    assert coeffs.shape == (101, 8) and off.max() + 263 <= X.shape[1]
    P = off.shape[0]
    power = np.empty(P, np.float64)
    oracle().oracle_das_fir8_f64(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index), index.size,
                                 _p32(coeffs), power.ctypes.data_as(_f64p))
    return power


def reference_fir_table():
    """The reference's filter_coeffs[101][8], parsed from its header WHERE IT LIES (nothing is copied
    into the repo); None when the reference tree is absent (GPU box)."""
    import re

    path = REFERENCE_TREE / "src/dsp/filter.h"
    if not path.exists():
        return None
    nums = re.findall(r"-?\d+\.\d+(?:[eE][-+]?\d+)?", path.read_text().split("filter_coeffs")[1])
    table = np.array(nums[: 101 * 8], dtype=np.float32).reshape(101, 8)
    return table


def ref_fir_table_probe():
    """The weights the reference's compiled 8-tap delay() (oracle/_ref/libref_das_fir.so) applies, recovered
    by running it on a unit impulse for every fraction k/100: [101, 8].  Works wherever oracle/_ref is
    (also on the GPU box, where the reference tree is not); None when the library is absent."""
    if not ref_available("fir"):
        return None
    lib = ref("fir")
    impulse = np.zeros(263, np.float32)
    impulse[7] = 1.0
    table = np.empty((101, 8), np.float32)
    for k in range(101):
        out = np.zeros(256, np.float32)
        lib.ref_delay(_p32(out), _p32(impulse), float(np.float32(k / 100.0)))
        table[k] = out[7::-1]
    return table


def delay_fir8(out, signal, fraction, coeffs):
    """The restated 8-tap delay() (delay.cpp:31-40): accumulates into `out` [256] in place."""
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    signal = np.ascontiguousarray(signal, np.float32)
    assert out.dtype == np.float32 and out.shape == (256,) and signal.size >= 263 and coeffs.shape == (101, 8)
    oracle().oracle_delay_fir8(_p32(out), _p32(signal), float(fraction), _p32(coeffs))
    return out


def ref_bench(X, off, frac, index=None, min_seconds=1.0, variant="avx2"):
    """frames/s of the reference-kernel loop nest on one thread over the P pixels of `off` (variant "fir": the
    build without -mavx2, whose delay() is the 8-tap table variant)."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    P = off.shape[0]
    power = np.empty(P, np.float32)
    done = C.c_int(0)
    fps = ref(variant).ref_das_bench(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1],
                                    _pi(index), index.size, _p32(power), float(min_seconds), C.byref(done))
    return fps, done.value


def ref_bench_mt(X, off, frac, threads, index=None, min_seconds=1.0, variant="avx2"):
    """frames/s of the same loop nest with the pixels dealt to `threads` host threads (the reference itself
    runs one); also returns the power it computed, for a check against the one-thread result."""
    X, off, frac, index = _sweep_args(X, off, frac, index)
    P = off.shape[0]
    power = np.empty(P, np.float32)
    done = C.c_int(0)
    fps = ref(variant).ref_das_bench_mt(_p32(X), X.shape[1], _pi(off), _p32(frac), P, off.shape[1], _pi(index),
                                       index.size, _p32(power), float(min_seconds), int(threads), C.byref(done))
    return fps, done.value, power


def heatmap_u8(power) -> np.ndarray:
    power = np.ascontiguousarray(power, np.float32)
    pix = np.empty(power.shape, np.uint8)
    oracle().oracle_heatmap_u8(_p32(power), power.size, pix.ctypes.data_as(_u8p))
    return pix


def resize_linear_u8(pix, out_rows: int, out_cols: int) -> np.ndarray:
    pix = np.ascontiguousarray(pix, np.uint8)
    out = np.empty((out_rows, out_cols), np.uint8)
    rc = oracle().oracle_resize_linear_u8(pix.ctypes.data_as(_u8p), pix.shape[0], pix.shape[1],
                                          out.ctypes.data_as(_u8p), out_rows, out_cols)
    if rc != 0:
        raise ValueError("oracle_resize_linear_u8: upscaling only")
    return out


def calibrate(X, reference_power_level=1e-5):
    X = np.ascontiguousarray(X, np.float32)
    assert X.shape[0] == 64
    index = np.empty(64, np.int32)
    corr = np.empty(64, np.float32)
    med = C.c_float(0)
    n = oracle().oracle_calibrate(_p32(X), X.shape[1], reference_power_level, _pi(index), _p32(corr),
                                  C.byref(med))
    return index[:n].copy(), corr[:n].copy(), med.value


def unpack_exposure(stream, n_sensors) -> np.ndarray:
    stream = np.ascontiguousarray(stream, np.int32)
    assert stream.shape[0] == 256
    block = np.empty((n_sensors, 256), np.float32)
    oracle().oracle_unpack_exposure(_pi(stream), stream.shape[1], n_sensors, _p32(block))
    return block
