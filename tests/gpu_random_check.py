"""Randomised parity sweep, run as a child process by tests/test_gpu_random.py so that the kernel
shape can be forced through the environment (the library reads its tuning knobs once per process).

Random, NOT geometry-derived tables: arbitrary integer offsets inside the legal range, arbitrary
fractions, arbitrary active-mic subsets, ragged pixel counts, odd batches, short histories."""
import importlib
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import util  # noqa: E402
from oracle import oracle_py  # noqa: E402

pkg = importlib.import_module("beamforming-lk_amd")
import os  # noqa: E402

MATH = pkg.MATH_F32_EXACT if os.environ.get("AWPU_TEST_MATH") == "exact" else pkg.MATH_F32_FAST
# AWPU_TEST_PATH=device: frames and power stay in device memory (awpu_hip_process_device reads the full
# [batch][streams][hist] layout in place; the host entry uploads a compacted window instead), each case
# also swept as two pixel shards and, where the wire format allows (hist 1024, <= 256 streams), from the ring
DEVICE_PATH = os.environ.get("AWPU_TEST_PATH") == "device"
# AWPU_TEST_INTERP=fir8: the 8-tap table variant of delay() (reads 6 samples further); AWPU_TEST_REUSE=1:
# every case re-targets ONE handle per geometry (new table, new mic list, gains on and off) instead of a
# fresh handle, so stale device tables or caches would show
FIR8 = os.environ.get("AWPU_TEST_INTERP") == "fir8"
REUSE = os.environ.get("AWPU_TEST_REUSE") == "1"


def main(seed: int, cases: int) -> int:
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(cases):
        n_streams = int(rng.choice([64, 128, 192, 256, 512]))
        lut_stride = n_streams + int(rng.choice([0, 0, 7]))
        hist = int(rng.choice([600, 777, 1024]))
        P = int(rng.choice([1, 5, 63, 64, 65, 130, 257]))
        grid_columns = 0
        if os.environ.get("AWPU_TEST_GRID") == "1":  # a real rows x cols grid, row length passed as the hint
            cols = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 100]))
            P = cols * int(rng.choice([1, 2, 3, 5, 8]))
            grid_columns = cols
        batch = int(rng.integers(1, 8))
        if os.environ.get("AWPU_TEST_BATCH") == "1":  # one frame per call in every case (the single-frame kernels)
            batch = 1
        usable = int(rng.integers(1, n_streams + 1))
        reach = 257 + (6 if FIR8 else 0)
        spread = int(rng.choice([0, 3, 40, 120, hist - reach]))  # window width control
        base = int(rng.integers(0, hist - reach - spread + 1))
        off = rng.integers(base, base + spread + 1, size=(P, lut_stride)).astype(np.int32)
        frac = rng.uniform(0, 1, size=(P, lut_stride)).astype(np.float32)
        frac[rng.uniform(size=frac.shape) < 0.05] = 0.0
        if grid_columns and os.environ.get("AWPU_TEST_COINCIDE") == "1":
            # like a real table: the integer delay of a mic changes slowly down a grid column (a step of one sample
            # now and then), so most vertical neighbours coincide -- what the shared-read and shared-sum blocks feed on
            rows_ = P // grid_columns
            steps = rng.choice([-1, 0, 0, 0, 0, 0, 1], size=(rows_, grid_columns, lut_stride))
            steps[0] = 0
            walk = off.reshape(rows_, grid_columns, lut_stride)[:1] + np.cumsum(steps, axis=0)
            off = np.clip(walk, base, base + spread).reshape(P, lut_stride).astype(np.int32)
        index = rng.permutation(n_streams)[:usable].astype(np.int32)
        X = util.hash_frames(n_streams, hist, seed=1000 + case, batch=batch)
        if FIR8:
            table = util.synthetic_fir_table()
            with pkg.Engine(math=pkg.MATH_F32_FAST, n_pixels=P, n_streams=n_streams, lut_stride=lut_stride, hist=hist, max_batch=batch,
                            interp=pkg.binding.INTERP_FIR8) as eng:
                eng.set_delay_table(off, frac)
                eng.set_active_mics(index)
                eng.set_fir_table(table)
                gains = rng.uniform(0.5, 2.0, n_streams).astype(np.float32) if case % 3 == 1 else None
                eng.set_mic_gains(gains)
                power = eng.process(X)
            for b in range(batch):
                Xb = X[b] * gains[:, None] if gains is not None else X[b]
                err = util.power_rel_err_unfloored(power[b], oracle_py.das_fir8_f32(Xb, off, frac, table, index))
                worst = max(worst, err)
                if not err < util.POWER_RTOL:
                    print(f"FAIL case {case} (fir8): streams {n_streams} hist {hist} P {P} batch {batch} usable {usable} "
                          f"window {spread + reach} frame {b}: rel err {err:.3e}")
                    return 1
            continue
        if REUSE:
            with pkg.Engine(n_pixels=P, n_streams=n_streams, lut_stride=lut_stride, hist=hist, max_batch=batch, math=MATH) as eng:
                for turn in range(3):  # same handle: other table, other mics, gains on / off
                    off_t = rng.integers(base, base + spread + 1, size=(P, lut_stride)).astype(np.int32)
                    frac_t = rng.uniform(0, 1, size=(P, lut_stride)).astype(np.float32)
                    index_t = rng.permutation(n_streams)[: int(rng.integers(1, n_streams + 1))].astype(np.int32)
                    gains = rng.uniform(0.5, 2.0, n_streams).astype(np.float32) if turn == 1 else None
                    eng.set_delay_table(off_t, frac_t)
                    eng.set_active_mics(index_t)
                    eng.set_mic_gains(gains)
                    power = eng.process(X)
                    for b in range(batch):
                        Xb = X[b] * gains[:, None] if gains is not None else X[b]
                        err = util.power_rel_err_unfloored(power[b], oracle_py.das_f32(Xb, off_t, frac_t, index_t))
                        worst = max(worst, err)
                        if not err < util.POWER_RTOL:
                            print(f"FAIL case {case} turn {turn} (reuse): streams {n_streams} hist {hist} P {P} batch {batch} "
                                  f"usable {index_t.size} frame {b}: rel err {err:.3e}")
                            return 1
            continue
        eng = pkg.Engine(n_pixels=P, n_streams=n_streams, lut_stride=lut_stride, hist=hist, max_batch=batch, math=MATH,
                         grid_columns=grid_columns)
        with eng:
            eng.set_delay_table(off, frac)
            eng.set_active_mics(index)
            if grid_columns and case % 3 == 1:  # per-mic gains on a third of the grid cases
                gains = rng.uniform(0.5, 2.0, n_streams).astype(np.float32)
                eng.set_mic_gains(gains)
                X = X * gains[None, :, None]  # what the oracle sees; the engine gets the unscaled frames
                X_in = X / gains[None, :, None]
                X_in = util.hash_frames(n_streams, hist, seed=1000 + case, batch=batch)
            else:
                X_in = X
            if DEVICE_PATH:
                import torch

                d_X = torch.from_numpy(X_in).cuda()
                d_P = torch.zeros((batch, P), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()  # (torch fills on its own stream; the engine's stream does not wait for it)
                eng.process_device(d_X.data_ptr(), batch, d_P.data_ptr())
                eng.synchronize()
                power = d_P.cpu().numpy()
            else:
                power = eng.process(X_in)
        if DEVICE_PATH and P > 1:  # two ragged pixel shards must tile the single-handle result bit for bit
            cut = int(rng.integers(1, P))
            parts = []
            for a, b in ((0, cut), (cut, P)):
                with pkg.Engine(n_pixels=P, n_streams=n_streams, lut_stride=lut_stride, hist=hist, max_batch=batch,
                                math=MATH, pixel_begin=a, pixel_count=b - a) as sh:
                    sh.set_delay_table(off[a:b], frac[a:b])
                    sh.set_active_mics(index)
                    parts.append(sh.process(X))
            if not np.array_equal(np.concatenate(parts, axis=1), power) and util.power_rel_err(np.concatenate(parts, axis=1), power) > 2e-6:
                print(f"FAIL case {case}: shards [0,{cut}) + [{cut},{P}) differ from the whole grid")
                return 1
        for b in range(batch):
            want = oracle_py.das_f32(X[b], off, frac, index)
            err = util.power_rel_err_unfloored(power[b], want)
            worst = max(worst, err)
            if not err < util.POWER_RTOL:
                print(f"FAIL case {case}: streams {n_streams} stride {lut_stride} hist {hist} P {P} batch {batch} "
                      f"usable {usable} window {spread + 257} frame {b}: rel err {err:.3e}")
                return 1
    print(f"OK {cases} cases, worst rel err {worst:.2e}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), int(sys.argv[2])))
