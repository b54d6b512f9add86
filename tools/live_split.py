#!/usr/bin/env python3
"""Where a live block's time goes through the C ABI: awpu_hip_ingest_block alone and awpu_hip_process_ring alone, with
and without the grid's row length given (tools/live_rate.py times the two together)."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for name in (sys.argv[1:] or ["c1", "c2", "headline"]):
    spec = S.WORKLOADS[name]
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz)
    rng = np.random.default_rng(0)
    msg = np.zeros(256, np.dtype([("h", "u1", (8,)), ("stream", "<i4", (256,))]))
    msg["stream"] = rng.integers(-(1 << 20), 1 << 20, (256, 256), dtype=np.int32)
    wire = msg.tobytes()
    for cols in (spec.res, 0, spec.res, 0):
        with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, grid_columns=cols) as eng:
            eng.set_delay_table(off, frac)
            eng.set_active_mics(None)
            for _ in range(5):
                eng.ingest_block(wire)
                eng.process_ring()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                eng.ingest_block(wire)
            eng.synchronize()
            t_ing = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                eng.process_ring()
            t_ring = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                eng.ingest_block(wire)
                eng.process_ring()
            t_both = (time.perf_counter() - t0) / n
            print(f"{name} grid_columns={cols}: ingest {t_ing * 1e6:.1f} us, process_ring {t_ring * 1e6:.1f} us, both {t_both * 1e6:.1f} us")
