"""Generate tests/golden/*.npz from the REFERENCE's own compiled delay() (oracle/_ref).

Run in the authoring container (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden.py

What is stored is data only: table inputs (off, frac, index), seeds of the exact integer-hash
frames (tests/util.hash_frames) and the outputs the reference kernel produced for them.
Expected values come from oracle/_ref/libref_das_avx2.so, i.e. /root/reference/src/dsp/delay.cpp
compiled unmodified with the reference's flags (oracle/Makefile), driven through the loop nest of
src/dsp/mimo.cpp:121-151 (oracle/ref_mimo_driver.cpp).  Table inputs come from the oracle's
restatement of src/dsp/mimo.cpp:20-59.
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from oracle import oracle_py as O  # noqa: E402
import util  # noqa: E402

_f32p = C.POINTER(C.c_float)


def delay_kat():
    """Known answers of delay(out, signal, fraction), src/dsp/delay.cpp:16-26."""
    ref = O.ref("avx2")
    assert ref.ref_variant() == 1
    sig = util.hash_frames(1, 300, seed=11, scale=1.0)[0, 0]
    acc0 = util.hash_frames(1, 256, seed=12, scale=4.0)[0, 0]
    fractions = np.array([0.0, 1.0, 0.5, 0.25, 0.999999, 1e-7, 0.3333333, 0.75, 0.6180339], np.float32)
    starts = np.array([0, 1, 7, 43, 0, 13, 2, 30, 5], np.int32)
    outs = np.empty((fractions.size, 256), np.float32)
    for j, (f, s) in enumerate(zip(fractions, starts)):
        out = acc0.copy()
        window = np.ascontiguousarray(sig[s:s + 257])
        ref.ref_delay(out.ctypes.data_as(_f32p), window.ctypes.data_as(_f32p), float(f))
        outs[j] = out
    np.savez_compressed(HERE / "delay_kat.npz", sig_seed=11, acc_seed=12, fractions=fractions,
                        starts=starts, expected=outs)
    print("delay_kat.npz", outs.shape)


def delay_kat_fir8():
    """Known answers of the reference's OTHER delay(), the 8-tap table variant its build selects without
    -mavx2 (src/dsp/delay.cpp:31-40), from oracle/_ref/libref_das_fir.so:
      * impulse responses: signal = a unit impulse at sample 7, fraction k/100 for k = 0..100; out[7 - t] is
        then exactly the weight the compiled reference puts on tap t (one product by 1.0, the rest by 0.0);
      * a noise signal through a few fractions, on a non-zero accumulator.
    Only outputs of the compiled function are stored."""
    ref = O.ref("fir")
    assert ref.ref_variant() == 2
    impulse = np.zeros(263, np.float32)
    impulse[7] = 1.0
    fractions = (np.arange(101) / 100.0).astype(np.float32)
    rows = (fractions * np.float32(100.0) + np.float32(0.5)).astype(np.int32)  # delay.cpp:32-33
    assert np.array_equal(rows, np.arange(101))
    response = np.empty((101, 8), np.float32)
    for k, f in enumerate(fractions):
        out = np.zeros(256, np.float32)
        ref.ref_delay(out.ctypes.data_as(_f32p), impulse.ctypes.data_as(_f32p), float(f))
        response[k] = out[7::-1]
        assert not out[8:].any()
    sig = util.hash_frames(1, 320, seed=21, scale=1.0)[0, 0]
    acc0 = util.hash_frames(1, 256, seed=22, scale=4.0)[0, 0]
    kat_f = np.array([0.0, 0.004, 0.005, 0.5, 0.25, 0.994, 0.995, 0.999999, 0.3333333, 0.6180339], np.float32)
    starts = np.array([0, 1, 7, 43, 0, 13, 2, 30, 5, 57], np.int32)
    outs = np.empty((kat_f.size, 256), np.float32)
    for j, (f, s0) in enumerate(zip(kat_f, starts)):
        out = acc0.copy()
        window = np.ascontiguousarray(sig[s0:s0 + 263])
        ref.ref_delay(out.ctypes.data_as(_f32p), window.ctypes.data_as(_f32p), float(f))
        outs[j] = out
    np.savez_compressed(HERE / "delay_kat_fir8.npz", impulse_at=7, impulse_fractions=fractions, impulse_response=response,
                        sig_seed=21, acc_seed=22, fractions=kat_f, starts=starts, expected=outs)
    print("delay_kat_fir8.npz", response.shape, outs.shape)
    return response


def sweep_case_fir8(name, arrays_x, arrays_y, res, pixels, seed, index=None, hist=1024, fov=180.0):
    """The sweep of mimo.cpp:121-151 around the reference's 8-tap delay() (libref_das_fir.so, which carries
    the reference's coefficient table inside): inputs and the powers it produced."""
    xyz = O.create_tiled_antenna(arrays_x, arrays_y)
    off, frac = O.compute_delay_lut(xyz, res, res, fov)
    off, frac = off[pixels], frac[pixels]
    n = xyz.shape[1]
    X = util.hash_frames(n, hist, seed=seed)[0]
    power = O.das_fir8_f32(X, off, frac, np.zeros((101, 8), np.float32), index, impl="ref")
    np.savez_compressed(
        HERE / f"{name}.npz", arrays=np.array([arrays_x, arrays_y]), res=res, fov=fov, pixels=pixels,
        seed=seed, hist=hist, off=off, frac=frac,
        index=np.arange(n, dtype=np.int32) if index is None else index.astype(np.int32), power=power)
    print(f"{name}.npz", off.shape, "power range", power.min(), power.max())


def sweep_case(name, arrays_x, arrays_y, res, pixels, seed, index=None, hist=1024, fov=180.0):
    xyz = O.create_tiled_antenna(arrays_x, arrays_y)
    off, frac = O.compute_delay_lut(xyz, res, res, fov)
    off, frac = off[pixels], frac[pixels]
    n = xyz.shape[1]
    X = util.hash_frames(n, hist, seed=seed)[0]
    power, out = O.das_f32(X, off, frac, index, want_out=True, impl="ref")
    np.savez_compressed(
        HERE / f"{name}.npz", arrays=np.array([arrays_x, arrays_y]), res=res, fov=fov, pixels=pixels,
        seed=seed, hist=hist, off=off, frac=frac,
        index=np.arange(n, dtype=np.int32) if index is None else index.astype(np.int32),
        power=power, out_first=out[:4], out_last=out[-1:])
    print(f"{name}.npz", off.shape, "power range", power.min(), power.max())


DC_OFFSETS = np.array([1e-4, 1e-3, 1e-2, 0.25], np.float32)


def sweep_case_dc(name, arrays_x, arrays_y, res, pixels, seed, hist=1024, fov=180.0):
    """The same sweep on DC-BIASED frames: X = hash_frames(seed) + offset (one fp32 add per sample), for every offset in
    DC_OFFSETS -- an ADC bias, which src/fpga/pipeline.cpp:290 does not remove and the reference's moving average
    (mimo.cpp:131-134) exists to cancel.  The reference sums the bias 256-fold into out[] before its stencil removes it,
    so its fp32 result carries rounding noise that grows with the offset; an implementation is "within 1e-5 of the
    reference" on such input only if it keeps the reference's operation order (AWPU_MATH_F32_EXACT).  Stored: powers and
    the first / last pixel's out[] from the reference's compiled delay(), per offset."""
    xyz = O.create_tiled_antenna(arrays_x, arrays_y)
    off, frac = O.compute_delay_lut(xyz, res, res, fov)
    off, frac = off[pixels], frac[pixels]
    n = xyz.shape[1]
    X0 = util.hash_frames(n, hist, seed=seed)[0]
    power = np.empty((DC_OFFSETS.size, len(pixels)), np.float32)
    out_first = np.empty((DC_OFFSETS.size, 4, 256), np.float32)
    out_last = np.empty((DC_OFFSETS.size, 1, 256), np.float32)
    for k, dc in enumerate(DC_OFFSETS):
        X = (X0 + dc).astype(np.float32)
        power[k], out = O.das_f32(X, off, frac, None, want_out=True, impl="ref")
        out_first[k], out_last[k] = out[:4], out[-1:]
    np.savez_compressed(HERE / f"{name}.npz", arrays=np.array([arrays_x, arrays_y]), res=res, fov=fov, pixels=pixels,
                        seed=seed, hist=hist, off=off, frac=frac, index=np.arange(n, dtype=np.int32), offsets=DC_OFFSETS,
                        power=power, out_first=out_first, out_last=out_last)
    print(f"{name}.npz", off.shape, "power range per offset", power.min(axis=1), power.max(axis=1))


def steer_split(xyz, theta, phi):
    """Particle::steer, src/dsp/particle.cpp:37-49, on the restated steering vector."""
    off = np.empty((len(theta), xyz.shape[1]), np.int32)
    frac = np.empty(off.shape, np.float32)
    for d, (t, p) in enumerate(zip(theta, phi)):
        tau = O.steering_delays_f32(xyz, float(t), float(p)).astype(np.float64)
        whole = np.trunc(tau)
        frac[d] = (tau - whole).astype(np.float32)
        off[d] = 256 - whole.astype(np.int32)
    return off, frac


def beams_case(name, seed, index=None):
    """Particle::beam / Particle::das (particle.cpp:51-103) around the reference delay(): a monopulse
    quadruple around each of a few directions, as GradientParticle::step asks for them."""
    xyz = O.create_antenna()
    rng = np.random.Generator(np.random.PCG64(seed))
    centre_t = rng.uniform(0.05, 1.2, 6)
    centre_p = rng.uniform(-np.pi, np.pi, 6)
    theta = np.concatenate([[t - 0.05, t + 0.05, t, t] for t in centre_t] + [[0.0, np.pi / 2]])
    phi = np.concatenate([[p, p, p - 0.05, p + 0.05] for p in centre_p] + [[0.0, 3.0]])
    off, frac = steer_split(xyz, theta, phi)
    X = util.hash_frames(64, 1024, seed=seed)[0]
    power, beams = O.particle_beams(X, off, frac, index, impl="ref")
    np.savez_compressed(HERE / f"{name}.npz", seed=seed, theta=theta, phi=phi, off=off, frac=frac,
                        index=np.arange(64, dtype=np.int32) if index is None else index.astype(np.int32),
                        power=power, beams=beams)
    print(f"{name}.npz", beams.shape, "power range", power.min(), power.max())


def main():
    O.build(ref=True)
    delay_kat()
    beams_case("beams_c1", seed=106)
    beams_case("beams_c1_ragged", seed=107, index=np.array([s for s in range(64) if s % 9 != 4], np.int32))
    rng = np.random.Generator(np.random.PCG64(2024))
    # c1 geometry (one 8x8 array, 32x32 grid, fov 180): every 7th pixel, all 64 mics
    sweep_case("sweep_c1", 1, 1, 32, np.arange(0, 1024, 7), seed=101)
    # ragged active-mic list as calibrate() would leave it: 51 of 64 mics, not in id order at the ends
    keep = np.sort(rng.choice(64, size=51, replace=False)).astype(np.int32)
    sweep_case("sweep_c1_ragged", 1, 1, 32, np.arange(3, 1024, 19), seed=102, index=keep)
    # a single active mic (degenerate) and fov 90
    sweep_case("sweep_c1_onemic", 1, 1, 32, np.arange(0, 1024, 97), seed=103, index=np.array([37]), fov=90.0)
    # 256 mics (4 arrays side by side), 128x128 grid: 48 pixels incl. the corners (theta clipped)
    pix = np.unique(np.concatenate([[0, 127, 16256, 16383], rng.choice(16384, 44, replace=False)]))
    sweep_case("sweep_headline", 4, 1, 128, pix, seed=104, hist=640)
    # 512 mics (4x2 arrays), 128x128 grid, shortest legal history
    pix = np.unique(np.concatenate([[0, 16383], rng.choice(16384, 22, replace=False)]))
    sweep_case("sweep_c3", 4, 2, 128, pix, seed=105, hist=520)
    # ---- DC-biased frames (round 4): the same geometries with an offset on every sample
    dc_rng = np.random.Generator(np.random.PCG64(2025))  # (its own stream: the draws above and below stay what they were)
    sweep_case_dc("sweep_c1_dc", 1, 1, 32, np.arange(0, 1024, 7), seed=121)
    pix = np.unique(np.concatenate([[0, 127, 16256, 16383], dc_rng.choice(16384, 44, replace=False)]))
    sweep_case_dc("sweep_headline_dc", 4, 1, 128, pix, seed=124, hist=640)
    # ---- the 8-tap variant of delay() (the reference built without -mavx2)
    delay_kat_fir8()
    sweep_case_fir8("sweep_c1_fir8", 1, 1, 32, np.arange(1, 1024, 9), seed=111)
    keep = np.sort(rng.choice(64, size=47, replace=False)).astype(np.int32)
    sweep_case_fir8("sweep_c1_ragged_fir8", 1, 1, 32, np.arange(5, 1024, 23), seed=112, index=keep)
    pix = np.unique(np.concatenate([[0, 127, 16256, 16383], rng.choice(16384, 36, replace=False)]))
    sweep_case_fir8("sweep_headline_fir8", 4, 1, 128, pix, seed=113, hist=640)
    pix = np.unique(np.concatenate([[0, 16383], rng.choice(16384, 18, replace=False)]))
    sweep_case_fir8("sweep_c3_fir8", 4, 2, 128, pix, seed=114, hist=528)


if __name__ == "__main__":
    main()
