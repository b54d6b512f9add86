#!/usr/bin/env python3
"""slab_times.py -- sweep time of every rank's slab of an N-rank run, measured one after the other on ONE GPU.

    python tools/slab_times.py [--workload headline] [--ranks 8] [--batch 1024] [--reps 4] [--weighted]

The N-GPU step takes as long as its slowest rank, and the slabs are not alike: rows near the edge of the sine-space
grid see other integer-delay statistics than rows in the middle (the quad shape's cost is 20 + 8 x differing pixels
per quad and mic).  Prints per slab: rows, the table's quad cost (computed here from the reference-format table) and
the kernel time; --interleaved deals row groups of four round-robin (sharding.shard_rows_interleaved).
"""
from __future__ import annotations

import argparse
import importlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def quad_cost(off: np.ndarray, cols: int) -> float:
    rows = off.shape[0] // cols
    o = off.reshape(rows, cols, -1)[: rows // 4 * 4].reshape(rows // 4, 4, cols, -1)
    ref = o[:, 1]
    d0, d2, d3 = o[:, 0] != ref, o[:, 2] != ref, o[:, 3] != ref
    together = d2 & (o[:, 2] == o[:, 3])
    return float(20.0 + (8.0 * (d0.sum() + d2.sum() + d3.sum()) - 4.0 * together.sum()) / d0.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="headline")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--interleaved", action="store_true", help="row groups of four dealt round-robin (bench.py's default for N > 1)")
    ap.add_argument("--packed", action="store_true", help="sweep packed frame pairs (no pack pass in the timed launch)")
    ap.add_argument("--only", default="", help="comma-separated ranks to time (default all)")
    args = ap.parse_args()
    pkg = importlib.import_module("beamforming-lk_amd")
    sharding = importlib.import_module("beamforming-lk_amd.sharding")
    S = pkg.synthetic
    spec = S.WORKLOADS[args.workload]
    xyz = S.geometry(spec)
    dev = torch.device("cuda", 0)
    distinct = S.make_frames(xyz, 64, seed=1234)
    d_full = torch.from_numpy(distinct).to(dev).repeat((args.batch + 63) // 64, 1, 1)[: args.batch].contiguous()
    stream = torch.cuda.Stream(device=dev)
    shards = sharding.all_shards(spec.res, spec.res, args.ranks, interleaved=args.interleaved)
    off_all, _ = S.delay_table(spec, xyz)
    window = (int(off_all.min()), int(off_all.max()) + 257)
    only = [int(x) for x in args.only.split(",")] if args.only else list(range(args.ranks))
    rows = []
    for shard in shards:
        if shard.rank not in only:
            continue
        off, frac = S.delay_table_for(spec, xyz, shard.row_ranges)
        d_power = torch.zeros((args.batch, shard.pixel_count), dtype=torch.float32, device=dev)
        with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=args.batch, pixel_begin=shard.pixel_begin,
                        pixel_count=shard.pixel_count, grid_columns=spec.res, window=window if args.packed else None) as eng:
            eng.set_delay_table(off, frac)
            eng.set_active_mics(None)
            if args.packed:
                d_pk = torch.zeros(eng.packed_bytes(args.batch) // 4, dtype=torch.float32, device=dev)
                eng.pack_frames(d_full.data_ptr(), args.batch, d_pk.data_ptr())
                eng.synchronize()
                run = lambda: eng.process_packed(d_pk.data_ptr(), args.batch, d_power.data_ptr(), stream.cuda_stream)
            else:
                run = lambda: eng.process_device(d_full.data_ptr(), args.batch, d_power.data_ptr(), stream.cuda_stream)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                for _ in range(3):
                    run()
                ev[0].record(stream)
                for _ in range(args.reps):
                    run()
                ev[1].record(stream)
                torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / args.reps
        rec = {"rank": shard.rank, "row_begin": shard.row_begin, "rows": shard.row_count,
               "quad_cost": round(quad_cost(off, spec.res), 2) if shard.row_count % 4 == 0 else None, "kernel_ms": round(ms, 4)}
        if args.interleaved:
            rec["row_ranges"] = [list(r) for r in shard.row_ranges]
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    ms = [r["kernel_ms"] for r in rows]
    print(json.dumps({"max_ms": max(ms), "mean_ms": float(np.mean(ms)), "imbalance": max(ms) / float(np.mean(ms))}))


if __name__ == "__main__":
    main()
