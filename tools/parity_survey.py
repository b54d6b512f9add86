#!/usr/bin/env python3
"""Full-grid, unfloored parity survey of every sweep shape against the oracle (GPU box only).

    python tools/parity_survey.py [--workloads c1,c2,headline,c3,c4slab] [--out gpurun_out/parity_survey.json]

For each workload: the bench's own frame (plane wave + noise, PCG64 seed 1234) and a hash-noise frame, through
  batch  (frame pairs: quad / pair / stationary shape; with and without the row-length hint),
  single (one frame per call: quad1 / round-1 shapes; with and without the hint),
  exact  (the reference-order kernels: one frame without the hint -- das_exact_pair_kernel -- and a batch with it -- das_exact_quad_kernel),
compared on EVERY pixel with oracle.das_f32 (the reference's operations) and oracle.das_f64.
Prints one JSON record per case: tests/util.parity_report.  Test infrastructure: uses oracle/.
"""
from __future__ import annotations

import argparse
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

import util  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c1,c2,headline,c3,c4slab")
    ap.add_argument("--out", default="")
    ap.add_argument("--interp", default="lerp", choices=["lerp", "fir8"], help="fir8: the 8-tap variant of delay() (batch and single shapes; a Blackman-windowed-sinc table)")
    args = ap.parse_args()
    fir = args.interp == "fir8"
    table = util.synthetic_fir_table() if fir else None
    pkg = importlib.import_module("beamforming-lk_amd")
    sharding = importlib.import_module("beamforming-lk_amd.sharding")
    from oracle import oracle_py

    S = pkg.synthetic
    records = []
    for wl in args.workloads.split(","):
        slab = wl == "c4slab"
        spec = S.WORKLOADS["c4" if slab else wl]
        xyz = S.geometry(spec)
        if slab:
            shard = sharding.shard_rows(spec.res, spec.res, 8, 3)
            off, frac = S.delay_table(spec, xyz, shard.row_begin, shard.row_count)
            begin, count = shard.pixel_begin, shard.pixel_count
        else:
            off, frac = S.delay_table(spec, xyz)
            begin, count = 0, spec.n_pixels
        frames = np.concatenate([S.make_frames(xyz, 3, seed=1234),
                                 util.hash_frames(spec.n_mics, 1024, seed=11, batch=1)])
        refs = []
        t0 = time.perf_counter()
        for b in (0, 3):
            if fir:
                refs.append((b, oracle_py.das_fir8_f32(frames[b], off, frac, table), oracle_py.das_fir8_f64(frames[b], off, frac, table)))
            else:
                refs.append((b, oracle_py.das_f32(frames[b], off, frac), oracle_py.das_f64(frames[b], off, frac)))
        t_or = time.perf_counter() - t0
        for mode, math, batch, hint in (("batch", "fast", 4, True), ("batch", "fast", 4, False),
                                        ("single", "fast", 1, True), ("single", "fast", 1, False),
                                        ("exact", "exact", 1, False), ("exact", "exact", 4, True)):
            eng = pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=batch,
                             math=pkg.MATH_F32_FAST if math == "fast" else pkg.MATH_F32_EXACT,
                             interp=pkg.binding.INTERP_FIR8 if fir else pkg.binding.INTERP_LERP,
                             pixel_begin=begin, pixel_count=count, grid_columns=spec.res if hint else 0)
            with eng:
                eng.set_delay_table(off, frac)
                eng.set_active_mics(None)
                if fir:
                    eng.set_fir_table(table)
                for b, r32, r64 in refs:
                    if batch == 1:
                        got = eng.process(frames[b:b + 1])[0]
                    else:  # the frame under test sits in a batch of four (odd and even slots of a pair)
                        order = [b, (b + 1) % 4, (b + 2) % 4, b]
                        power = eng.process(frames[order])
                        got = power[0]
                        assert np.array_equal(power[3], power[0]), "the same frame in another slot of a pair differs"
                    rep = util.parity_report(got, r32, r64)
                    rep.update({"workload": wl, "interp": args.interp, "mode": mode, "grid_columns": hint, "batch": batch,
                                "kernel": pkg.binding.KERNEL_NAMES[eng.stats().kernel_variant],
                                "frame": "plane wave + noise" if b == 0 else "hash noise"})
                    records.append(rep)
                    print(json.dumps(rep), flush=True)
        print(f"# {wl}: oracle f32+f64 on {count} pixels x 2 frames took {t_or:.1f} s", flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(records, indent=1))
    worst = max(records, key=lambda r: r["max_rel_unfloored"])
    print("# worst case (the bound is 1e-5, flat):", json.dumps(worst))
    sys.exit(0 if all(r["ok"] for r in records) else 1)


if __name__ == "__main__":
    main()
