"""Quick GPU check of the quad shape against the oracle (development aid; the tests cover the same ground).
    AWPU_FAST_QUADS=1 python tools/quad_check.py"""
import importlib
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import util  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
for wl, rows, batch in [("c1", (0, 32), 2), ("c1", (3, 6), 3), ("c2", (16, 8), 4), ("headline", (40, 8), 4), ("c3", (100, 12), 2),
                        ("c1", (0, 32), 1), ("c1", (3, 6), 1), ("c2", (0, 64), 1), ("headline", (40, 30), 1), ("c3", (100, 12), 1)]:
    spec = S.WORKLOADS[wl]
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz, *rows)
    frames = S.make_frames(xyz, batch, seed=5)
    eng = pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=batch, pixel_begin=rows[0] * spec.res,
                     pixel_count=rows[1] * spec.res, grid_columns=spec.res)
    with eng:
        eng.set_delay_table(off, frac)
        eng.set_active_mics(None)
        power = eng.process(frames)
        st = eng.stats()
    worst = 0.0
    for b in range(batch):
        want = O.das_f32(frames[b], off, frac)
        worst = max(worst, util.power_rel_err(power[b], want))
    print(f"{wl} rows {rows} batch {batch}: max rel err {worst:.3e}  kernel {st.last_kernel_ms:.3f} ms", flush=True)
    if not worst < 1e-5:
        bad = np.argwhere(np.abs(power[0] - O.das_f32(frames[0], off, frac)) / O.das_f32(frames[0], off, frac).max() > 1e-5)
        print("  bad pixels (row, col):", [(int(p) // spec.res, int(p) % spec.res) for p in bad[:12, 0]], "of", len(bad))
