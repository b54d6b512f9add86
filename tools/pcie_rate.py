#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point awpu_hip_process (pageable numpy memory):
H2D of the frames + sweep + D2H of the power, per call.  Noted in DESIGN.md; never the bench value."""
import importlib
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
spec = S.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "headline"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
xyz = S.geometry(spec)
off, frac = S.delay_table(spec, xyz)
frames = S.make_frames(xyz, B)
for math, name in ((pkg.MATH_F32_FAST, "fast"), (pkg.MATH_F32_EXACT, "exact")):
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=B, math=math) as eng:
        eng.set_delay_table(off, frac)
        eng.set_active_mics(None)
        eng.process(frames)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            eng.process(frames)
        dt = (time.perf_counter() - t0) / n
        st = eng.stats()
        print(f"{spec.name} math={name} batch {B}: host-buffer call {dt * 1e3:.2f} ms -> {B / dt:.0f} frames/s "
              f"(kernel alone {st.last_kernel_ms:.2f} ms -> {B / st.last_kernel_ms * 1e3:.0f} frames/s)")
