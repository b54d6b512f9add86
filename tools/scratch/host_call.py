import importlib, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for name in ("ref_default", "c2"):
    spec = S.WORKLOADS[name]
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz)
    frames = S.make_frames(xyz, 2, seed=1)
    for mode in ("exact", "fast"):
        math = pkg.MATH_F32_EXACT if mode == "exact" else pkg.MATH_F32_FAST
        with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=1, grid_columns=spec.res, math=math) as eng:
            eng.set_delay_table(off, frac); eng.set_active_mics(None)
            for _ in range(5): eng.process(frames[:1])
            t0 = time.perf_counter()
            for _ in range(200): p = eng.process(frames[:1])
            dt = (time.perf_counter() - t0) / 200 * 1e6
            print(f"{name} [{mode}] {pkg.binding.KERNEL_NAMES[eng.stats().kernel_variant]}: {dt:.1f} us per synchronous awpu_hip_process call (python ctypes included); last_kernel_ms {eng.stats().last_kernel_ms*1e3:.1f} us")
