import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
pkg = importlib.import_module("beamforming-lk_amd")
oracle = importlib.import_module("oracle.oracle_py")
S = pkg.synthetic
xyz = pkg.create_antenna()
rows = cols = 100
off, frac = pkg.build_delay_table(xyz, rows, cols, 180.0)
frames = (S.make_frames(xyz, 3, seed=55) + np.float32(0.125)).astype(np.float32)
P = rows * cols
names = pkg.binding.KERNEL_NAMES
for ragged in (False, True):
    for use_gain in (False, True):
        index = np.array([k for k in range(64) if k % 5 != 2], np.int32) if ragged else None
        gains = (0.5 + np.arange(64) / 64.0).astype(np.float32) if use_gain else None
        X = frames[1] * gains[:, None] if use_gain else frames[1]
        want_p, want_out = oracle.das_f32(X, off, frac, index=index, want_out=True)
        with pkg.Engine(n_pixels=P, n_streams=64, math=pkg.MATH_F32_EXACT, max_batch=3, grid_columns=cols) as eng:
            eng.set_delay_table(off, frac)
            eng.set_active_mics(index)
            if use_gain:
                eng.set_mic_gains(gains)
            for B, sl in ((1, slice(1, 2)), (3, slice(0, 3))):
                d_X = torch.from_numpy(frames[sl].copy()).cuda()
                d_P = torch.empty((B, P), dtype=torch.float32, device="cuda")
                d_S = torch.full((B, P, 256), float("nan"), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                eng.process_device_sums(d_X.data_ptr(), B, d_P.data_ptr(), d_S.data_ptr())
                eng.synchronize()
                k = 0 if B == 1 else 1
                sums = d_S.cpu().numpy()[k]
                bad = np.argwhere(sums != want_out)
                print(f"ragged={ragged} gain={use_gain} B={B} kernel={names[eng.stats().kernel_variant]} sums mismatches={len(bad)} first={bad[:3].tolist()} "
                      f"power maxrel={np.max(np.abs(d_P.cpu().numpy()[k]-want_p)/want_p):.2e}")
