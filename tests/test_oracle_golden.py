"""The CPU oracle against (a) the committed golden vectors, which were produced by the
reference's own compiled delay() (tests/golden/make_golden.py), (b) that reference build
itself when oracle/_ref is present, (c) an fp64 restatement, (d) the physics known answer."""
from pathlib import Path

import numpy as np
import pytest

import util

GOLDEN = Path(__file__).resolve().parent / "golden"
SWEEPS = ["sweep_c1", "sweep_c1_ragged", "sweep_c1_onemic", "sweep_headline", "sweep_c3"]


def test_delay_known_answers_bit_exact(oracle):
    """delay(), src/dsp/delay.cpp:16-26: the restatement reproduces the reference bit for bit."""
    import ctypes as C

    g = np.load(GOLDEN / "delay_kat.npz")
    sig = util.hash_frames(1, 300, seed=int(g["sig_seed"]), scale=1.0)[0, 0]
    acc0 = util.hash_frames(1, 256, seed=int(g["acc_seed"]), scale=4.0)[0, 0]
    f32p = C.POINTER(C.c_float)
    for f, s, want in zip(g["fractions"], g["starts"], g["expected"]):
        out = acc0.copy()
        window = np.ascontiguousarray(sig[s:s + 257])
        oracle.oracle().oracle_delay_lerp(out.ctypes.data_as(f32p), window.ctypes.data_as(f32p), float(f))
        assert np.array_equal(out, want), f"fraction {f}"


@pytest.mark.parametrize("name", SWEEPS)
def test_sweep_golden(oracle, name):
    """MIMOWorker::update, src/dsp/mimo.cpp:121-151: pre-epilogue sums bit-exact, power to 2e-6."""
    g = np.load(GOLDEN / f"{name}.npz")
    ax, ay = g["arrays"]
    X = util.hash_frames(64 * int(ax) * int(ay), int(g["hist"]), seed=int(g["seed"]))[0]
    power, out = oracle.das_f32(X, g["off"], g["frac"], g["index"], want_out=True)
    assert np.array_equal(out[:4], g["out_first"])
    assert np.array_equal(out[-1:], g["out_last"])
    # the reference build is -Ofast: its epilogue sum order is the compiler's; power agrees to rounding
    assert util.power_rel_err(power, g["power"]) < 2e-6


@pytest.mark.parametrize("name", ["sweep_c1_dc", "sweep_headline_dc"])
def test_sweep_golden_dc_biased(oracle, name):
    """The same on DC-BIASED frames (hash frames + {1e-4 .. 0.25}; round 4): the restatement's pre-epilogue sums are the
    reference object code's bit for bit at every offset; powers agree to the sum-order noise of the -Ofast epilogue
    (2e-6, as without the bias: measured 7e-7 at every offset).  Also records how far
    the reference's own fp32 result is from exact (fp64) sums on this input -- the noise an implementation that
    re-orders the arithmetic cannot reproduce."""
    g = np.load(GOLDEN / f"{name}.npz")
    ax, ay = g["arrays"]
    X0 = util.hash_frames(64 * int(ax) * int(ay), int(g["hist"]), seed=int(g["seed"]))[0]
    for k, dc in enumerate(g["offsets"]):
        X = (X0 + np.float32(dc)).astype(np.float32)
        power, out = oracle.das_f32(X, g["off"], g["frac"], g["index"], want_out=True)
        assert np.array_equal(out[:4], g["out_first"][k]) and np.array_equal(out[-1:], g["out_last"][k])
        assert util.power_rel_err_unfloored(power, g["power"][k]) < 2e-6
        p64 = oracle.das_f64(X, g["off"], g["frac"], g["index"])
        print(f"{name} offset {dc}: oracle vs reference build {util.power_rel_err_unfloored(power, g['power'][k]):.2e}, "
              f"reference build vs exact {util.power_rel_err_unfloored(g['power'][k], p64):.2e}")


@pytest.mark.parametrize("name", ["beams_c1", "beams_c1_ragged"])
def test_particle_beams_golden(oracle, pkg, name):
    """Particle::beam / Particle::das, src/dsp/particle.cpp:51-103: the beams the reference's
    delay() produced are reproduced bit for bit, the powers to rounding (its -Ofast epilogue order is
    the compiler's); the stored tables are what the library's Particle::steer mirror gives."""
    g = np.load(GOLDEN / f"{name}.npz")
    X = util.hash_frames(64, 1024, seed=int(g["seed"]))[0]
    power, beams = oracle.particle_beams(X, g["off"], g["frac"], g["index"])
    assert np.array_equal(beams, g["beams"])
    assert util.power_rel_err(power, g["power"]) < 2e-6
    off, frac = pkg.steer_table(oracle.create_antenna(), g["theta"], g["phi"])
    assert np.array_equal(off, g["off"]) and np.array_equal(frac, g["frac"])
    if oracle.ref_available():
        p_r, b_r = oracle.particle_beams(X, g["off"], g["frac"], g["index"], impl="ref")
        assert np.array_equal(b_r, g["beams"]) and np.array_equal(p_r, g["power"])


@pytest.mark.parametrize("name", SWEEPS)
def test_golden_tables_match_lut_restatement(oracle, name):
    """The stored tables are what the restated computeDelayLUT (mimo.cpp:20-59) gives today."""
    g = np.load(GOLDEN / f"{name}.npz")
    ax, ay = (int(v) for v in g["arrays"])
    xyz = oracle.create_tiled_antenna(ax, ay)
    off, frac = oracle.compute_delay_lut(xyz, int(g["res"]), int(g["res"]), float(g["fov"]))
    assert np.array_equal(off[g["pixels"]], g["off"])
    assert np.array_equal(frac[g["pixels"]], g["frac"])


def test_oracle_matches_reference_build_live(oracle):
    """Same seeded plane-wave frame through the restatement and through oracle/_ref."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    xyz = oracle.create_antenna()
    off, frac = oracle.compute_delay_lut(xyz, 24, 24)
    X = util.hash_frames(64, 1024, seed=5)[0]
    p_o, out_o = oracle.das_f32(X, off, frac, want_out=True)
    p_r, out_r = oracle.das_f32(X, off, frac, want_out=True, impl="ref")
    assert np.array_equal(out_o, out_r)
    assert util.power_rel_err(p_o, p_r) < 2e-6


def test_f32_sweep_close_to_f64(oracle):
    xyz = oracle.create_tiled_antenna(4, 1)
    off, frac = oracle.compute_delay_lut(xyz, 16, 16)
    X = util.hash_frames(256, 640, seed=9)[0]
    p32 = oracle.das_f32(X, off, frac)
    p64 = oracle.das_f64(X, off, frac)
    assert util.power_rel_err(p32, p64) < 5e-6


@pytest.mark.parametrize("arrays,res", [((1, 1), 32), ((4, 1), 32), ((4, 2), 24)])
def test_lut_against_fp64(oracle, arrays, res):
    """fp32 table (mimo.cpp:46-54) vs the fp64 closed form: recombined delay within 5e-5 samples (a few fp32 ulps at tau ~ 100),
    fractions in [0,1), offsets inside the history, one zero-delay mic per pixel."""
    xyz = oracle.create_tiled_antenna(*arrays)
    off, frac = oracle.compute_delay_lut(xyz, res, res)
    tau64 = oracle.compute_delays_f64(xyz, res, res)
    tau32 = (256 - off) + frac.astype(np.float64)
    assert np.abs(tau32 - tau64).max() < 5e-5
    assert frac.min() >= 0.0 and frac.max() < 1.0
    assert off.max() == 256 and off.min() >= 0
    assert np.all((tau32 == 0).sum(axis=1) >= 1)


def test_tau_max_of_baseline_geometries(oracle):
    """8x8 -> tau_max 28, 32x8 -> 91, 32x16 -> 98 samples (the survey rounds to ~99)."""
    for arrays, want in [((1, 1), 28), ((4, 1), 91), ((4, 2), 98)]:
        xyz = oracle.create_tiled_antenna(*arrays)
        off, _ = oracle.compute_delay_lut(xyz, 32, 32)
        assert 256 - off.min() == want


def test_tiled_antenna_1x1_is_create_antenna(oracle):
    assert np.array_equal(oracle.create_tiled_antenna(1, 1), oracle.create_antenna(8, 8, 0.02))
    xyz = oracle.create_antenna()
    # antenna.cpp:66-73 at 8x8: x = c*0.02 - 0.07, y = r*0.02 - 0.07
    assert np.allclose(xyz[0, :8], np.arange(8) * 0.02 - 0.07, atol=1e-7)
    assert np.allclose(xyz[1, ::8], np.arange(8) * 0.02 - 0.07, atol=1e-7)
    assert np.all(xyz[2] == 0)


def test_plane_wave_peaks_at_source_pixel(oracle, pkg):
    """Physics known answer (pipeline.cpp:105-135 signal model): the heatmap maximum is the
    pixel that looks at the source."""
    S = pkg.synthetic
    spec = S.WORKLOADS["c1"]
    xyz = oracle.create_tiled_antenna(spec.arrays_x, spec.arrays_y)
    off, frac = oracle.compute_delay_lut(xyz, spec.res, spec.res, spec.fov)
    for theta_deg, phi_deg in [(20.0, 35.0), (0.5, 0.0), (35.0, -120.0)]:
        th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
        X = S.make_frames(xyz, 1, seed=3, theta=th, phi=ph)[0]
        power = oracle.das_f32(X, off, frac)
        r, c = divmod(int(power.argmax()), spec.res)
        er, ec = S.source_pixel(spec, th, ph)
        assert abs(r - er) <= 1 and abs(c - ec) <= 1, (theta_deg, phi_deg, (r, c), (er, ec))


def test_heatmap_u8(oracle):
    p = np.array([0.0, 1e-6, 5e-6, 1e-5], np.float32)
    pix = oracle.heatmap_u8(p)
    assert pix.tolist() == [0, 25, 127, 255]


def test_resize_linear_u8_known_answers(oracle):
    """cv::resize INTER_LINEAR on 8-bit (aw_processing_unit.cpp:252).  OpenCV is absent here (parity
    unpinned for this function): the expected 2x2 -> 4x4 image below was worked out by hand from
    OpenCV's fixed-point recipe (weights 2048/1536/512, ((b*(S>>4))>>16 summed, +2, >>2)."""
    src = np.array([[0, 100], [200, 40]], np.uint8)
    want = [[0, 25, 75, 100], [50, 59, 76, 85], [150, 126, 79, 55], [200, 160, 80, 40]]
    assert oracle.resize_linear_u8(src, 4, 4).tolist() == want
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear_u8(img, 16, 16), img)              # same size: identity
    assert np.all(oracle.resize_linear_u8(np.full((7, 9), 201, np.uint8), 40, 31) == 201)  # constants survive
    big = oracle.resize_linear_u8(img, 64, 64)
    assert big.min() >= img.min() and big.max() <= img.max()                      # a convex combination
    assert np.array_equal(big[[0, -1]][:, [0, -1]], img[[0, -1]][:, [0, -1]])      # corners replicate
    ramp = np.tile(np.arange(0, 256, 8, dtype=np.uint8), (4, 1))
    up = oracle.resize_linear_u8(ramp, 4, 128).astype(int)
    assert np.all(np.diff(up, axis=1) >= 0)                                       # monotone stays monotone
    with pytest.raises(ValueError):
        oracle.resize_linear_u8(img, 8, 8)


def test_calibrate_restatement(oracle):
    """aw_processing_unit.cpp:128-200: a dead mic and a loud mic are dropped."""
    X = util.hash_frames(64, 1024, seed=21, scale=2.0 ** -7)[0].copy()
    X[5] = 0.0      # dead: power < median * 1e-3
    X[17] *= 64.0   # far too loud: |power - median| > 1e-4
    index, corr, median = oracle.calibrate(X)
    assert 5 not in index and 17 not in index and index.size == 62
    pw = (X.astype(np.float64) ** 2).mean(axis=1)
    assert np.allclose(corr, 1e-5 / pw[index], rtol=1e-4)


def test_unpack_exposure(oracle):
    """pipeline.cpp:277-290: every other group of 8 columns is mirrored, scaled by 2^-23."""
    stream = np.arange(256 * 256, dtype=np.int32).reshape(256, 256) - 30000
    block = oracle.unpack_exposure(stream, 128)
    # first group of 8 is "inverted" (inverted toggles to 1 at sensor 0)
    assert block[0, 0] == np.float32(stream[0, 7]) / np.float32(8388608.0)
    assert block[7, 3] == np.float32(stream[3, 0]) / np.float32(8388608.0)
    assert block[8, 0] == np.float32(stream[0, 8]) / np.float32(8388608.0)
    assert block[16, 255] == np.float32(stream[255, 23]) / np.float32(8388608.0)


def test_fir8_restatement_matches_reference_build_live(oracle):
    """delay() FIR variant, src/dsp/delay.cpp:31-40: the reference's own non-AVX2 build (which
    carries its filter.h table) against the restatement fed the same table, parsed from the
    reference header where it lies."""
    table = oracle.reference_fir_table()
    if table is None or not oracle.ref_available("fir"):
        pytest.skip("reference tree / oracle/_ref FIR build not available")
    assert table.shape == (101, 8) and abs(float(table[0, 3]) - 1.0) < 1e-4
    xyz = oracle.create_antenna()
    off, frac = oracle.compute_delay_lut(xyz, 10, 10)
    X = util.hash_frames(64, 1024, seed=44)[0]
    p_o = oracle.das_fir8_f32(X, off, frac, table)
    p_r = oracle.das_fir8_f32(X, off, frac, table, impl="ref")
    assert util.power_rel_err(p_o, p_r) < 5e-6


def test_fir8_at_zero_fraction_is_a_three_sample_delay(oracle):
    """Row 0 of a fractional-delay table is (numerically) a unit tap at index 3: the FIR sweep then
    equals the linear sweep of the same frame read 3 samples later with fraction 0 -> weight on s[i+1],
    i.e. offsets + 2."""
    table = np.zeros((101, 8), np.float32)
    table[:, 3] = 1.0
    xyz = oracle.create_antenna()
    off, _ = oracle.compute_delay_lut(xyz, 6, 6)
    frac = np.zeros_like(off, dtype=np.float32)
    X = util.hash_frames(64, 1024, seed=45)[0]
    p_fir = oracle.das_fir8_f32(X, off, frac, table)
    p_lin = oracle.das_f32(X, off + 2, frac)
    assert np.allclose(p_fir, p_lin, rtol=1e-6, atol=0)


FIR8_SWEEPS = ["sweep_c1_fir8", "sweep_c1_ragged_fir8", "sweep_headline_fir8", "sweep_c3_fir8"]


def fir8_table_from_fixture():
    """The weights the reference's compiled 8-tap delay() applies, as measured from it (impulse responses,
    tests/golden/delay_kat_fir8.npz): what the GPU box, which has no reference tree, feeds set_fir_table."""
    g = np.load(GOLDEN / "delay_kat_fir8.npz")
    table = g["impulse_response"]
    assert table.shape == (101, 8) and abs(float(table[0, 3]) - 1.0) < 1e-4
    return g, table


def test_fir8_known_answers(oracle):
    """delay(), 8-tap variant (src/dsp/delay.cpp:31-40): the restatement fed the measured weights reproduces
    what the reference's non-AVX2 build returned for a noise signal (the reference is -Ofast: its 8-term
    sums are in the compiler's order, so a few ulp of the accumulator, not bits)."""
    g, table = fir8_table_from_fixture()
    sig = util.hash_frames(1, 320, seed=int(g["sig_seed"]), scale=1.0)[0, 0]
    acc0 = util.hash_frames(1, 256, seed=int(g["acc_seed"]), scale=4.0)[0, 0]
    for f, s0, want in zip(g["fractions"], g["starts"], g["expected"]):
        out = oracle.delay_fir8(acc0.copy(), sig[s0:s0 + 263], float(f), table)
        assert np.abs(out - want).max() <= 4 * np.spacing(np.float32(np.abs(want).max()))
    # the rounding of the row index (delay.cpp:32-33): 0.004 -> row 0, 0.005 -> row 1, 0.994 -> 99, 0.995 -> 100
    rows = (g["fractions"] * np.float32(100.0) + np.float32(0.5)).astype(np.int32)
    assert rows[:3].tolist() == [0, 0, 1] and rows[5:8].tolist() == [99, 100, 100]


def test_fir8_fixture_table_is_what_the_build_applies(oracle):
    """Where oracle/_ref is present the impulse responses are re-measured live and must equal the fixture; where
    the reference tree is present too, they must equal its filter.h, read where it lies."""
    _, table = fir8_table_from_fixture()
    live = oracle.ref_fir_table_probe()
    if live is None:
        pytest.skip("oracle/_ref FIR build not available")
    assert np.array_equal(live, table)
    header = oracle.reference_fir_table()
    if header is not None:
        assert np.array_equal(header, table)


@pytest.mark.parametrize("name", FIR8_SWEEPS)
def test_fir8_sweep_goldens(oracle, name):
    """The restated FIR8 sweep against powers produced by the reference's own non-AVX2 delay() in the loop
    nest of mimo.cpp:121-151 (tests/golden/make_golden.py, libref_das_fir.so)."""
    _, table = fir8_table_from_fixture()
    g = np.load(GOLDEN / f"{name}.npz")
    ax, ay = g["arrays"]
    X = util.hash_frames(64 * int(ax) * int(ay), int(g["hist"]), seed=int(g["seed"]))[0]
    power = oracle.das_fir8_f32(X, g["off"], g["frac"], table, g["index"])
    assert util.power_rel_err(power, g["power"]) < 5e-6


def test_bf16_accumulator_restatement(oracle):
    """The checker of the AWPU_MATH_BF16_ACC mode (BASELINE configs[4]: bf16 vs fp32 accumulator): keeping the
    running sums in bfloat16 moves the per-pixel power by about a percent -- three orders of magnitude outside
    the 1e-5 the fp32 path is held to -- and never by nothing."""
    xyz = oracle.create_antenna()
    off, frac = oracle.compute_delay_lut(xyz, 16, 16)
    X = util.hash_frames(64, 1024, seed=61)[0]
    p32 = oracle.das_f32(X, off, frac)
    p16 = oracle.das_bf16acc(X, off, frac)
    err = util.power_rel_err(p16, p32)
    assert 1e-4 < err < 1e-1
    # one mic: the sum is one term, rounded once -- within bf16's half ulp (2^-9) of the fp32 sum, per sample
    one = np.array([11], np.int32)
    assert util.power_rel_err(oracle.das_bf16acc(X, off, frac, one), oracle.das_f32(X, off, frac, one)) < 2.0 ** -7


def test_parity_report_is_unfloored():
    """tests/util.parity_report states the parity claim in the north star's wording (per-pixel, relative, NO floor):
    a pixel 1e-9 of the frame peak that is off by 2e-5 relative must fail it although the floored metric passes."""
    import util

    ref = np.array([1.0, 0.5, 1e-9, 2e-5], np.float64)
    good = ref * (1 + 4e-6)
    rep = util.parity_report(good, ref, ref * (1 + 1e-6))
    assert rep["ok"] and rep["pixels_below_floor"] == 2 and rep["pixels_over_1e5"] == 0 and rep["bound"] == 1e-5
    bad = good.copy()
    bad[2] = ref[2] * (1 + 2e-5)
    rep = util.parity_report(bad, ref, ref * (1 + 1e-6))
    assert not rep["ok"] and rep["pixels_over_1e5"] == 1 and abs(rep["max_rel_unfloored"] - 2e-5) < 1e-9
    assert util.power_rel_err(bad, ref) < 1e-5  # ... which the floored metric would have let through
    # where the reference's own fp32 is far from exact sums, `ok` stays strict (1e-5 flat against the reference's fp32
    # result); only the separately NAMED allowance follows the reference's distance to exact: 3 x that distance
    rep = util.parity_report(bad, ref, ref * (1 + 8e-6))
    assert rep["bound"] == 1e-5 and not rep["ok"]
    assert abs(rep["noise_bound"] - 2.4e-5) < 1e-9 and rep["ok_within_reference_noise"]
    assert util.power_rel_err_unfloored(np.array([0.0, 1.0]), np.array([0.0, 1.0])) == 0.0
    assert util.power_rel_err_unfloored(np.array([1e-30, 1.0]), np.array([0.0, 1.0])) == np.inf


def test_fir8_f64_tiebreaker_is_close_to_the_f32_restatement(oracle):
    import util

    xyz = oracle.create_antenna()
    off, frac = oracle.compute_delay_lut(xyz, 8, 8)
    off = np.minimum(off, 1024 - 263).astype(np.int32)
    X = util.hash_frames(64, 1024, seed=2)[0]
    table = util.synthetic_fir_table()
    p32 = oracle.das_fir8_f32(X, off, frac, table)
    p64 = oracle.das_fir8_f64(X, off, frac, table)
    assert 0 < util.power_rel_err_unfloored(p32, p64) < 5e-6
