#!/usr/bin/env python3
"""20 000 synchronous one-frame host calls (awpu_hip_process, the reference's shipped shape, default mode) on changing frames, each result
compared bit for bit with the device-pointer path's: the completion flag of the sweep's last workgroup (das_exact_ndh_kernel<1, true>)
must never be seen before every power is in the pinned buffer.  Run under gpurun: python tools/host_call_stress.py"""
import importlib, sys, time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
spec = S.WORKLOADS["ref_default"]
xyz = S.geometry(spec)
off, frac = S.delay_table(spec, xyz)
frames = S.make_frames(xyz, 7, seed=5)
frames *= (1.0 + np.arange(7, dtype=np.float32))[:, None, None]
with pkg.Engine(n_pixels=spec.n_pixels, n_streams=64, math=(pkg.MATH_F32_FAST if "fast" in sys.argv else pkg.MATH_F32_EXACT), max_batch=1, grid_columns=spec.res) as eng:
    eng.set_delay_table(off, frac); eng.set_active_mics(None)
    want = []
    for k in range(7):
        d_X = torch.from_numpy(frames[k:k + 1].copy()).cuda(); d_P = torch.zeros((1, spec.n_pixels), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize(); eng.process_device(d_X.data_ptr(), 1, d_P.data_ptr()); eng.synchronize(); want.append(d_P.cpu().numpy()[0])
    bad = 0
    t0 = time.perf_counter()
    N = 20000
    for call in range(N):
        k = (3 * call) % 7
        got = eng.process(frames[k:k + 1])[0]
        bad += not np.array_equal(got, want[k])
    print("calls", N, "mismatches", bad, "us per call incl. the comparison", (time.perf_counter() - t0) / N * 1e6)
