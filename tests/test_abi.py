"""The C-ABI library on a box without a GPU: it loads, exports every symbol include/awpu_hip.h
declares, fails loudly instead of falling back to a CPU path, and its one-off host geometry
(create_antenna / steering_vector_spherical / computeDelayLUT mirrors) equals the oracle."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "awpu_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awpu_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.binding.load()
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/awpu_hip.h but not exported"
    assert sorted(pkg.binding.EXPORTED_SYMBOLS) == names
    assert lib.awpu_hip_abi_version() == 4


def test_shipping_library_reads_three_environment_variables(pkg):
    """Round-3 verdict, item 7: the kernel zoo's ~20 getenv switches are gone from the shipping library.  What is left:
    AWPU_SHAPE (tests force each production shape through the oracle), AWPU_LIVE_GRAPH and AWPU_GROUP_FORCE_COPY; the
    tuning variables of rounds 1-3 are read only by -DAWPU_TUNING_BUILD builds.  Checked on the binary itself."""
    import subprocess

    pkg.binding.load()
    out = subprocess.run(["strings", "-a", str(pkg._build.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    names = sorted({ln for ln in out.splitlines() if re.fullmatch(r"AWPU_[A-Z0-9_]+", ln)})
    assert names == ["AWPU_GROUP_FORCE_COPY", "AWPU_LIVE_GRAPH", "AWPU_SHAPE"], names


def test_shipping_build_has_no_wrong_result_switches(pkg):
    """AWPU_FAST_DEBUG's timing switches (no refill / no sweep / no tail pass / no barrier / register staging) give
    wrong heatmaps; they exist only in -DAWPU_TIMING_BUILD builds.  The default library exports no marker of such a
    build, its kernels test the constant 0 (das_kernels.h: AWPU_DBG), and the host masks the variable."""
    lib = pkg.binding.load()
    assert not hasattr(lib, "awpu_hip_timing_build"), "libawpu_hip.so was built with -DAWPU_TIMING_BUILD"
    header = (REPO / "beamforming-lk_amd" / "csrc" / "das_kernels.h").read_text()
    assert "#define AWPU_DBG(a, bit) 0" in header
    for name in ("das_fast.hip", "das_kernels.hip"):
        text = (REPO / "beamforming-lk_amd" / "csrc" / name).read_text()
        # every use of a wrong-result bit goes through the macro (bits 16, 256, 512, 4096 keep the results)
        for bit in (1, 2, 4, 8, 64):
            assert not re.search(rf"debug\s*&\s*{bit}\)", text), (name, bit)
    host = (REPO / "beamforming-lk_amd" / "csrc" / "awpu_hip.cpp").read_text()
    assert "debug &= awpu::kDebugSafeBits" in host


def test_cfg_struct_matches_header(pkg):
    cfg = pkg.binding.Cfg()
    pkg.binding.load().awpu_hip_default_cfg(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(pkg.binding.Cfg) == 96
    assert (cfg.n_streams, cfg.hist, cfg.lut_stride, cfg.max_batch) == (64, 1024, 64, 1)
    assert cfg.math == pkg.MATH_F32_EXACT and cfg.interp == 0  # the reference has one arithmetic: that one is the default


def test_no_cpu_fallback_without_device(pkg):
    """Without a gfx950 device creation fails with AWPU_ERR_NO_DEVICE -- never a CPU path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only boxes")
    with pytest.raises(pkg.AwpuError) as ei:
        pkg.Engine(n_pixels=16)
    assert ei.value.status == pkg.binding.ERR_NO_DEVICE


def test_device_table_builder_without_device(pkg):
    """awpu_hip_build_delay_table_device needs a gfx950 device and says so; it never falls back to the host builder."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only boxes")
    xyz = pkg.create_antenna()
    with pytest.raises(pkg.AwpuError) as ei:
        pkg.build_delay_table_device(xyz, 8, 8)
    assert ei.value.status == pkg.binding.ERR_NO_DEVICE
    lib = pkg.binding.load()
    assert lib.awpu_hip_build_delay_table_device(0, None, 64, 8, 8, 180.0, 0, 8, None, None) == pkg.binding.ERR_INVALID


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: no product source may reference it."""
    for path in (REPO / "beamforming-lk_amd").rglob("*"):
        if path.suffix in {".py", ".cpp", ".hip", ".h"}:
            text = path.read_text()
            assert "oracle_py" not in text and "das_oracle" not in text and "liboracle" not in text, path


def test_argument_errors(pkg):
    lib = pkg.binding.load()
    assert lib.awpu_hip_create(None, None) == pkg.binding.ERR_INVALID
    cfg = pkg.binding.Cfg()
    lib.awpu_hip_default_cfg(C.byref(cfg))
    h = C.c_void_p()
    assert lib.awpu_hip_create(C.byref(h), C.byref(cfg)) == pkg.binding.ERR_INVALID  # n_pixels 0
    cfg.n_pixels = 4
    cfg.struct_size = 8
    assert lib.awpu_hip_create(C.byref(h), C.byref(cfg)) == pkg.binding.ERR_INVALID
    assert b"struct_size" in lib.awpu_hip_last_error()
    assert lib.awpu_hip_destroy(None) == 0
    assert lib.awpu_hip_strerror(-2).startswith(b"no gfx950")


@pytest.mark.parametrize("arrays", [(1, 1), (4, 1), (4, 2)])
def test_host_geometry_equals_oracle(pkg, oracle, arrays):
    xyz = pkg.create_tiled_antenna(*arrays)
    assert np.array_equal(xyz, oracle.create_tiled_antenna(*arrays))
    for theta, phi in [(0.0, 0.0), (0.3, 1.1), (1.2, -2.5), (np.pi / 2, 0.7)]:
        assert np.array_equal(pkg.steering_delays(xyz, theta, phi), oracle.steering_delays_f32(xyz, theta, phi))


def test_create_antenna_equals_oracle(pkg, oracle):
    for cols, rows in [(8, 8), (32, 8), (5, 3)]:
        assert np.array_equal(pkg.create_antenna(cols, rows), oracle.create_antenna(cols, rows))


@pytest.mark.parametrize("arrays,res,fov", [((1, 1), 32, 180.0), ((4, 1), 20, 180.0), ((1, 1), 10, 90.0), ((4, 2), 12, 120.0)])
def test_delay_table_equals_oracle(pkg, oracle, arrays, res, fov):
    """awpu_hip_build_delay_table mirrors MIMOWorker::computeDelayLUT, src/dsp/mimo.cpp:20-59."""
    xyz = pkg.create_tiled_antenna(*arrays)
    off, frac = pkg.build_delay_table(xyz, res, res, fov)
    off_o, frac_o = oracle.compute_delay_lut(xyz, res, res, fov)
    assert np.array_equal(off, off_o) and np.array_equal(frac, frac_o)
    # row slabs (the per-rank shard of the multi-GPU path) tile the full table
    a = pkg.build_delay_table(xyz, res, res, fov, 0, res // 2)
    b = pkg.build_delay_table(xyz, res, res, fov, res // 2, res - res // 2)
    assert np.array_equal(np.concatenate([a[0], b[0]]), off) and np.array_equal(np.concatenate([a[1], b[1]]), frac)


def test_host_resize_equals_oracle(pkg, oracle):
    """awpu_hip_resize_linear_u8 (host, no device needed) against the restated cv::resize."""
    rng = np.random.default_rng(8)
    for (r, c, R, C_) in [(2, 2, 4, 4), (16, 16, 64, 64), (12, 20, 50, 33), (64, 64, 256, 256), (5, 1, 5, 9), (9, 9, 9, 9)]:
        img = rng.integers(0, 256, (r, c), dtype=np.uint8)
        assert np.array_equal(pkg.resize_linear_u8(img, R, C_), oracle.resize_linear_u8(img, R, C_)), (r, c, R, C_)
    with pytest.raises(Exception):
        pkg.resize_linear_u8(rng.integers(0, 256, (8, 8), dtype=np.uint8), 4, 8)


def test_heatmap_equals_oracle(pkg, oracle):
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 3e-5, 4096).astype(np.float32)
    assert np.array_equal(pkg.heatmap_u8(p), oracle.heatmap_u8(p))
    # an all-zero frame (0/0 in mimo.cpp:85; the reference's cast of the NaN is undefined): a black image
    zero = np.zeros(64, np.float32)
    assert not pkg.heatmap_u8(zero).any() and not oracle.heatmap_u8(zero).any()
    assert lib_last_error_of_null(pkg) == b""


def lib_last_error_of_null(pkg):
    return pkg.binding.load().awpu_hip_last_error_of(None)


def test_null_arguments_are_refused_not_dereferenced(pkg):
    """Every entry point of the C ABI with a null handle / null buffers: a negative status, no crash
    (no device needed: argument checks come first).  awpu_hip_destroy(NULL) is a no-op like free(NULL)."""
    B = pkg.binding
    lib = B.load()
    f, i, u = np.zeros(16, np.float32), np.zeros(16, np.int32), np.zeros(16, np.uint8)
    fp, ip, up = f.ctypes.data_as(B._f32p), i.ctypes.data_as(B._i32p), u.ctypes.data_as(B._u8p)
    seen = []

    def call(name, *args):
        status = getattr(lib, name)(*args)
        seen.append(name)
        assert status < 0, (name, status)

    assert lib.awpu_hip_destroy(None) == 0
    call("awpu_hip_set_delay_table", None, ip, fp)
    call("awpu_hip_set_active_mics", None, ip, 4)
    call("awpu_hip_set_fir_table", None, fp)
    call("awpu_hip_set_mic_gains", None, fp)
    call("awpu_hip_process", None, fp, 1, fp)
    call("awpu_hip_process_device", None, None, 1, None, None)
    call("awpu_hip_process_device_sums", None, None, 1, None, None, None)
    call("awpu_hip_process_async", None, fp, 1, fp)
    call("awpu_hip_wait", None)
    call("awpu_hip_synchronize", None)
    call("awpu_hip_packed_bytes", None, 2, None)
    call("awpu_hip_pack_frames", None, None, 2, None, None)
    call("awpu_hip_process_packed", None, None, 2, None, None)
    call("awpu_hip_ingest_block", None, None, 1032)
    call("awpu_hip_process_ring", None, fp)
    call("awpu_hip_ring_snapshot", None, fp)
    call("awpu_hip_live_block", None, None, 1032, fp, 4, 4, up, 8, 8, None, up)
    call("awpu_hip_heatmap_u8", None, 4, up)
    call("awpu_hip_heatmap_u8_device", None, None, 4, 1, None, 0, None, None)
    call("awpu_hip_upscale_u8_device", None, None, 4, 4, 1, None, None, 8, 8, None)
    call("awpu_hip_resize_linear_u8", None, 4, 4, up, 8, 8)
    call("awpu_hip_calibrate_device", None, None, 0, 1e-5, ip, fp, None, None, None)
    call("awpu_hip_calibrate_ring", None, 0, 1e-5, ip, fp, None, None)
    call("awpu_hip_calibrate_host", None, fp, 0, 1e-5, ip, fp, None, None)
    call("awpu_hip_beams", None, None, ip, fp, 1, fp, fp)
    call("awpu_hip_steer_table", None, 4, None, None, 1, ip, fp)
    call("awpu_hip_create_antenna", 8, 8, 0.02, None)
    call("awpu_hip_create_tiled_antenna", 0, 1, 0.02, fp)
    call("awpu_hip_steering_delays", None, 4, 0.1, 0.2, fp)
    call("awpu_hip_build_delay_table", None, 4, 2, 2, 180.0, 0, 2, ip, fp)
    call("awpu_hip_build_delay_table_device", 0, None, 4, 2, 2, 180.0, 0, 2, ip, fp)
    call("awpu_hip_get_stats", None, None)
    call("awpu_hip_group_peer_status", None, ip, 1)
    call("awpu_hip_create", None, None)
    untested = set(B.EXPORTED_SYMBOLS) - set(seen) - {"awpu_hip_destroy", "awpu_hip_default_cfg", "awpu_hip_strerror",
                                                      "awpu_hip_last_error", "awpu_hip_last_error_of",
                                                      "awpu_hip_abi_version"}
    assert not untested, untested
