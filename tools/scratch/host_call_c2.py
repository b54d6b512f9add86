import importlib, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for name in ("c2", "ref_default"):
  spec = S.WORKLOADS[name]
  xyz = S.geometry(spec)
  off, frac = S.delay_table(spec, xyz)
  frames = S.make_frames(xyz, 2, seed=1)
  for mode in (pkg.MATH_F32_EXACT, pkg.MATH_F32_FAST):
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=1, grid_columns=spec.res, math=mode) as eng:
        eng.set_delay_table(off, frac); eng.set_active_mics(None)
        for _ in range(5): eng.process(frames[:1])
        t0 = time.perf_counter()
        for _ in range(100): p = eng.process(frames[:1])
        a = (time.perf_counter() - t0) / 100 * 1e6
        d_X = torch.from_numpy(frames[:1]).cuda(); d_P = torch.zeros((1, spec.n_pixels), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(5): eng.process_device(d_X.data_ptr(), 1, d_P.data_ptr()); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): eng.process_device(d_X.data_ptr(), 1, d_P.data_ptr()); eng.synchronize()
        b = (time.perf_counter() - t0) / 100 * 1e6
        print(name, mode, pkg.binding.KERNEL_NAMES[eng.stats().kernel_variant], f"process {a:.1f} us | process_device+synchronize {b:.1f} us")
