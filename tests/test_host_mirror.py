"""The C++ host mirror of the reference's MIMO worker (beamforming-lk_amd/host): built with g++
against the C ABI, run as the reference's process would run it."""
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
EXE = REPO / "tests" / "host" / "test_mimo_worker"


EXACT = REPO / "tests" / "host" / "test_exact_signatures"


def build(pkg, oracle, exact=False):
    pkg.binding.load()  # builds libawpu_hip.so
    pkgdir = REPO / "beamforming-lk_amd"
    host = [str(pkgdir / "host" / f) for f in ("mimo_worker_hip.cpp", "aw_processing_unit_hip.cpp", "pipeline_hip.cpp",
                                                "aw_processing_unit.cpp")]
    # the exact-signature class needs cv::Mat: OpenCV is absent from this image, the tests bring a few-line stand-in
    extra = ["-DAWPU_WITH_OPENCV", f"-I{REPO / 'tests/host/mock_opencv'}"] if exact else []
    main, exe = ("test_exact_signatures.cpp", EXACT) if exact else ("test_mimo_worker.cpp", EXE)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", *extra, f"-I{REPO / 'include'}", f"-I{pkgdir / 'host'}",
           f"-I{REPO / 'oracle'}", str(REPO / "tests/host" / main), *host,
           f"-L{pkgdir}", "-lawpu_hip", f"-L{REPO / 'oracle'}", "-loracle_das", "-lm",
           f"-Wl,-rpath,{pkgdir}", f"-Wl,-rpath,{REPO / 'oracle'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_exact_signature_unit_compiles(pkg, oracle):
    """class AWProcessingUnit with the reference's own constructor and draw(cv::Mat*, cv::Mat*) signatures
    (aw_processing_unit.h:37-45,125) builds against the C ABI (AWPU_WITH_OPENCV)."""
    build(pkg, oracle, exact=True)
    assert EXACT.exists()


@pytest.mark.gpu
def test_exact_signature_unit_runs_like_the_control_unit_drives_it(pkg, oracle):
    build(pkg, oracle, exact=True)
    out = subprocess.run([str(EXACT)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_host_mirror_builds_and_fails_loudly_without_gpu(pkg, oracle):
    import torch

    build(pkg, oracle)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only boxes")
    out = subprocess.run([str(EXE), "--nogpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "OK nogpu" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_host_mirror_against_oracle(pkg, oracle):
    build(pkg, oracle)
    out = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def build_example(pkg):
    pkg.binding.load()
    exe = REPO / "examples" / "heatmap_min"
    pkgdir = REPO / "beamforming-lk_amd"
    subprocess.run(["gcc", "-O2", "-Wall", f"-I{REPO / 'include'}", str(REPO / "examples/heatmap_min.c"), f"-L{pkgdir}",
                    "-lawpu_hip", "-lm", f"-Wl,-rpath,{pkgdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)],
                   check=True, capture_output=True, text=True)
    return exe


def test_c_example_reports_the_missing_device(pkg):
    """examples/heatmap_min.c is plain C against include/awpu_hip.h; without a GPU it must say so (exit 2)."""
    import torch

    exe = build_example(pkg)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 2 and "no gfx950" in out.stderr, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_example_finds_the_source(pkg):
    out = subprocess.run([str(build_example(pkg))], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    S = pkg.synthetic
    er, ec = S.source_pixel(S.WORKLOADS["c1"])
    r, c = (int(v) for v in out.stdout.split("(")[1].split(")")[0].split(","))
    assert abs(r - er) <= 1 and abs(c - ec) <= 1 and "image value there 255" in out.stdout, out.stdout


def test_synthetic_noise_is_the_std_mt19937_stream(pkg, tmp_path):
    """SURVEY.md 8d seeds the synthetic inputs with std::mt19937(1234).  beamforming-lk_amd/synthetic.py draws its noise from
    numpy's legacy MT19937 seeding, which is the same generator: the first draws equal a g++-compiled std::mt19937's, and a
    frame's noise is NOISE * (2 r / 2^32 - 1) of consecutive draws (mic-major, sample-minor)."""
    src = tmp_path / "mt.cpp"
    src.write_text('#include <random>\n#include <cstdio>\nint main() { std::mt19937 g(1234); for (int i = 0; i < 6; i++) '
                   'std::printf("%u\\n", (unsigned) g()); }\n')
    subprocess.run(["g++", "-O1", "-o", str(tmp_path / "mt"), str(src)], check=True, capture_output=True)
    want = [int(v) for v in subprocess.run([str(tmp_path / "mt")], capture_output=True, text=True, check=True).stdout.split()]
    import numpy as np

    got = np.random.RandomState(1234).randint(0, 2 ** 32, size=6, dtype=np.uint32)
    assert [int(v) for v in got] == want
    S = pkg.synthetic
    xyz = S.geometry(S.WORKLOADS["c1"])
    frame = S.make_frames(xyz, 1, seed=1234)[0]
    clean = S.make_frames(xyz, 1, seed=1234)[0] - 0  # (deterministic)
    assert np.array_equal(frame, clean)
    # the noise of mic 0, samples 0..5, recovered by removing the plane wave generated with NOISE = 0
    noise = np.float64(S.NOISE) * (np.array(want, np.float64) * (2.0 / 4294967296.0) - 1.0)
    keep = S.NOISE
    try:
        S.NOISE = 0.0
        wave = S.make_frames(xyz, 1, seed=1234)[0]
    finally:
        S.NOISE = keep
    assert np.allclose(frame[0, :6].astype(np.float64) - wave[0, :6], noise, atol=2e-9)


def test_host_code_under_address_and_ub_sanitizers(pkg, oracle, tmp_path):
    """the reference has no sanitizer coverage; the host side here gets one on the CPU (GPU sanitizers are not
    available on the pool).  (a) csrc/geometry_host.cpp -- create_antenna, steer, computeDelayLUT -- compiled with
    -fsanitize=address,undefined beside a small driver; its table equals the oracle's.  (b) the C++ mirror classes
    (MIMOWorkerHip, AWProcessingUnitHip, PipelineHip) and tests/host/test_mimo_worker.cpp under the same flags, run on the
    no-device path (construction, the refused start, destruction order)."""
    import os

    import numpy as np

    pkg.binding.load()
    oracle.oracle()
    pkgdir = REPO / "beamforming-lk_amd"
    san = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-pthread"]
    main = tmp_path / "geometry_main.cpp"
    main.write_text(r'''
#include "awpu_hip.h"
#include <cstdio>
#include <vector>
int main() {
    std::vector<float> xyz(3 * 256);
    if (awpu_hip_create_tiled_antenna(4, 1, 0.02f, xyz.data()) != 0) return 1;
    const int res = 20;
    std::vector<int32_t> off((size_t) res * res * 256);
    std::vector<float> frac(off.size());
    if (awpu_hip_build_delay_table(xyz.data(), 256, res, res, 180.0f, 0, res, off.data(), frac.data()) != 0) return 2;
    FILE *f = std::fopen("table.bin", "wb");
    std::fwrite(off.data(), 4, off.size(), f);
    std::fwrite(frac.data(), 4, frac.size(), f);
    std::fclose(f);
    int32_t o1[64]; float f1[64]; double th = 0.4, ph = -1.3;
    std::vector<float> one(3 * 64);
    if (awpu_hip_create_antenna(8, 8, 0.02f, one.data()) != 0 || awpu_hip_steer_table(one.data(), 64, &th, &ph, 1, o1, f1) != 0) return 3;
    return awpu_hip_create_antenna(8, 8, 0.02f, nullptr) < 0 ? 0 : 4;  // (a null output is refused, not written through)
}
''')
    exe = tmp_path / "geometry_san"
    subprocess.run(["g++", *san, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{REPO / 'include'}", f"-I{pkgdir / 'csrc'}",
                    str(main), str(pkgdir / "csrc" / "geometry_host.cpp"), "-o", str(exe)], check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = np.fromfile(tmp_path / "table.bin", dtype=np.uint8)
    n = 20 * 20 * 256
    off_o, frac_o = oracle.compute_delay_lut(oracle.create_tiled_antenna(4, 1), 20, 20, 180.0)
    assert np.array_equal(raw[: 4 * n].view(np.int32).reshape(400, 256), off_o)
    assert np.array_equal(raw[4 * n:].view(np.float32).reshape(400, 256), frac_o)
    import torch

    if torch.cuda.is_available():
        return  # (the no-device path below is for CPU-only boxes; GPU sanitizers are not available on the pool)
    host = [str(pkgdir / "host" / f) for f in ("mimo_worker_hip.cpp", "aw_processing_unit_hip.cpp", "pipeline_hip.cpp", "aw_processing_unit.cpp")]
    exe2 = tmp_path / "mimo_worker_san"
    subprocess.run(["g++", *san, f"-I{REPO / 'include'}", f"-I{pkgdir / 'host'}", f"-I{REPO / 'oracle'}",
                    str(REPO / "tests/host/test_mimo_worker.cpp"), *host, f"-L{pkgdir}", "-lawpu_hip", f"-L{REPO / 'oracle'}", "-loracle_das",
                    "-lm", f"-Wl,-rpath,{pkgdir}", f"-Wl,-rpath,{REPO / 'oracle'}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe2)],
                   check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe2), "--nogpu"], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))  # (the HIP runtime's own start-up allocations are not ours to free)
    assert out.returncode == 0 and "OK nogpu" in out.stdout, out.stdout + out.stderr
