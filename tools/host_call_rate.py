#!/usr/bin/env python3
"""One frame per SYNCHRONOUS host call (awpu_hip_process on a pageable host frame: upload + sweep + read-back): the call
MIMOWorker::update makes once per 256-sample block (mimo.cpp:100-103), at the reference's shipped shape, both fp32 modes.
With a tuning build and AWPU_LIVE_TIMING=1 the library prints the call's breakdown (gather / upload / launch / wait / copy out)."""
import importlib, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
spec = S.WORKLOADS["ref_default"]
xyz = S.geometry(spec)
off, frac = S.delay_table(spec, xyz)
frames = S.make_frames(xyz, 8, seed=1)
for mode in (pkg.MATH_F32_EXACT, pkg.MATH_F32_FAST):
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=1, grid_columns=spec.res, math=mode) as eng:
        eng.set_delay_table(off, frac); eng.set_active_mics(None)
        for k in range(5): eng.process(frames[k % 8:k % 8 + 1])
        t0 = time.perf_counter()
        for k in range(100): p = eng.process(frames[k % 8:k % 8 + 1])
        print(mode, pkg.binding.KERNEL_NAMES[eng.stats().kernel_variant], (time.perf_counter() - t0) / 100 * 1e6, "us per call")
