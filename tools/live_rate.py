#!/usr/bin/env python3
"""Block rate of the live path through the C ABI: awpu_hip_ingest_block (256 raw wire datagrams, host ->
device ring, unpack on the GPU) + awpu_hip_process_ring (single-frame sweep of the ring's snapshot + D2H of
the power), per 256-sample block.  The array delivers 48828 / 256 = 190.7 blocks/s; this is how far above
real time one GPU runs the per-block work.  Noted in DESIGN.md; not the bench value."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for name in sys.argv[1:] or ["c1", "headline"]:
    spec = S.WORKLOADS[name]
    if spec.n_mics > 256:
        print(f"{spec.name}: the wire format carries at most 256 sensors per datagram, skipped")
        continue
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz)
    rng = np.random.default_rng(0)
    msg = np.zeros(256, np.dtype([("h", "u1", (8,)), ("stream", "<i4", (256,))]))
    msg["stream"] = rng.integers(-(1 << 20), 1 << 20, (256, 256), dtype=np.int32)
    wire = msg.tobytes()
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics) as eng:
        eng.set_delay_table(off, frac)
        eng.set_active_mics(None)
        for _ in range(5):
            eng.ingest_block(wire)
            eng.process_ring()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            eng.ingest_block(wire)
            eng.process_ring()
        dt = (time.perf_counter() - t0) / n
        print(f"{spec.name}: ingest + sweep + readback {dt * 1e3:.3f} ms per block -> {1 / dt:.0f} blocks/s "
              f"= {1 / dt / (48828 / 256):.0f} x real time")
        t0 = time.perf_counter()
        for _ in range(n):  # the whole display step in one call: + 8-bit heatmap + upscale to 1024 x 1024, images read back
            eng.live_block(wire, spec.res, spec.res, 1024, 1024, want_power=False)
        dt = (time.perf_counter() - t0) / n
        print(f"{spec.name}: live_block (block in, {spec.res}x{spec.res} and 1024x1024 images out) {dt * 1e3:.3f} ms per block "
              f"-> {1 / dt:.0f} blocks/s")
