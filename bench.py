#!/usr/bin/env python3
"""bench.py -- heatmap frames/s of the delay-and-sum sweep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline] [--batch B]

One "step" = one pass of the hot path over one batch of B synthetic frames already resident
in HBM: (N > 1: RCCL broadcast of the batch from rank 0, overlapped with the previous
step's sweep) + one sweep launch per rank over that rank's slab of the steering grid.
N > 1 is launched by torch.distributed.run, one rank per GPU; the grid (total work) is fixed,
so scaling is "strong".  With BENCH_ALT=1 and N > 1 a second, separately timed pass measures the other
decomposition (whole frames per rank, scatter instead of broadcast) and is reported as "alt_sharding"
beside the headline value (opt-in: it doubles the run and adds collectives that no multi-GPU box has
exercised yet, and the headline line must not depend on them).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak fp32 vector


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", help="c1 | c2 | headline | c3 | c4")
    ap.add_argument("--batch", type=int, default=128, help="frames per step (per sweep launch)")
    ap.add_argument("--math", default="fast", choices=["fast", "exact"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


def cpu_baseline(S, spec, xyz, off, frac, frame, seconds):
    """The reference's own delay() (oracle/_ref) in the loop nest of mimo.cpp:121-151, one
    thread, whole frames of the same workload until `seconds` have passed."""
    from oracle import oracle_py

    kind = "reference" if oracle_py.ref_available() else "port"
    if kind == "reference":
        fps, frames = oracle_py.ref_bench(frame, off, frac, None, min_seconds=seconds)
    else:
        t0 = time.perf_counter()
        frames = 0
        while True:
            oracle_py.das_f32(frame, off, frac)
            frames += 1
            if time.perf_counter() - t0 >= seconds:
                break
        fps = frames / (time.perf_counter() - t0)
    out = {
        "value": fps, "unit": "frames/s", "cores": 1, "kind": kind,
        "sample": f"{frames} whole frames of the same workload ({spec.name}), 1 thread, "
                  f"{os.cpu_count()} host cores present",
    }
    if kind == "reference":  # beside it: the same kernel with the pixels dealt to the box's CPU share
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()))
        fps_mt, frames_mt, _ = oracle_py.ref_bench_mt(frame, off, frac, threads, min_seconds=max(2.0, seconds / 4))
        out["multi_thread"] = {"value": fps_mt, "unit": "frames/s", "cores": threads,
                               "sample": f"{frames_mt} whole frames, pixels dealt to {threads} threads (the reference's "
                                         f"MIMO worker itself is one thread)"}
    return out


def measured_traffic(workload, batch, world):
    """HBM bytes per launch from the committed PMC run (profiles/r01_hbm_traffic.json, collected
    with tools/pmc_hbm.sh as MI355X_MICROARCH.md prescribes); None when it was not measured for
    this exact workload."""
    path = REPO / "profiles" / "r01_hbm_traffic.json"
    if world != 1 or not path.exists():
        return None
    rec = json.loads(path.read_text())
    if rec.get("workload") != workload:
        return None
    for m in rec["measurements"]:  # not proportional to the batch (L2 residency changes): exact matches only
        if m["frames_per_step"] == batch:
            return int(m["traffic_bytes_per_launch"])
    return None


def main():
    args = parse_args()
    # dmabuf IPC for RCCL between processes; read when the HSA runtime starts, so before torch touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sweep has no CPU path")
    # BENCH_REHEARSAL=1: run the N > 1 code path on a box with ONE GPU (all ranks on cuda:0, gloo
    # instead of RCCL) -- a correctness rehearsal of the sharding/broadcast logic, not a measurement.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module("beamforming-lk_amd")
    sharding = importlib.import_module("beamforming-lk_amd.sharding")
    S = pkg.synthetic
    spec = S.WORKLOADS[args.workload]
    B, K, W = args.batch, args.steps, args.warmup
    math = pkg.MATH_F32_FAST if args.math == "fast" else pkg.MATH_F32_EXACT

    # ---- one-off setup: geometry, this rank's slab of the delay table, frames in HBM
    shard = sharding.shard_rows(spec.res, spec.res, world, rank)
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz, shard.row_begin, shard.row_count)
    hist = pkg.binding.HIST
    win_begin = 0
    if world > 1:
        # Only the window [min off, max off + 257) of every mic is ever read, so that
        # is what travels: rank 0 cuts it out of its 1024-sample snapshots each step and broadcasts
        # [B][mics][Wc]; every rank sweeps with hist = Wc and offsets relative to the window.
        lo = torch.tensor([int(off.min())], dtype=torch.int64, device=dev)
        hi = torch.tensor([int(off.max())], dtype=torch.int64, device=dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        win_begin = int(lo.item())
        hist = ((int(hi.item()) - win_begin + 257 + 3) // 4) * 4 + 4
        off = off - win_begin
    eng = pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, hist=hist, math=math, max_batch=B,
                     device=local_rank, pixel_begin=shard.pixel_begin, pixel_count=shard.pixel_count,
                     grid_columns=0 if os.environ.get("BENCH_NO_GRID_HINT") else spec.res)
    eng.set_delay_table(off, frac)
    eng.set_active_mics(None)

    host_frames = S.make_frames(xyz, B, seed=args.seed) if rank == 0 else None
    d_full = torch.from_numpy(host_frames).to(dev) if rank == 0 else None  # the ingest layout [B][mics][1024]
    if world > 1:
        bufs = tuple(torch.zeros((B, spec.n_mics, hist), dtype=torch.float32, device=dev) for _ in range(2))
    else:
        bufs = (d_full, d_full)
    d_power = torch.zeros((B, shard.pixel_count), dtype=torch.float32, device=dev)
    bcast = sharding.FrameBroadcaster(bufs, src=0, mode=os.environ.get("BENCH_BCAST", "broadcast"))
    # a real (non-null) stream: its handle goes to the C ABI, and the torch events that time
    # the sweep are recorded on the same stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def post(k):
        if world > 1 and rank == 0:  # cut the window out of the resident snapshots (part of the step)
            bufs[k % 2].copy_(d_full[:, :, win_begin:win_begin + hist])
        bcast.post(k)

    def run_steps(n, ev=None):
        with torch.cuda.stream(stream):
            _run_steps(n, ev)

    def _run_steps(n, ev):
        post(0)
        for k in range(n):
            frames = bcast.wait(k)
            if k + 1 < n:
                post(k + 1)  # next batch travels while this one is swept
            if ev is not None:
                ev[0][k].record(stream)
            eng.process_device(frames.data_ptr(), B, d_power.data_ptr(), stream.cuda_stream)
            if ev is not None:
                ev[1][k].record(stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(W)
    fence()
    events = ([torch.cuda.Event(enable_timing=True) for _ in range(K)],
              [torch.cuda.Event(enable_timing=True) for _ in range(K)])
    t0 = time.perf_counter()
    run_steps(K, events)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(*events)]))  # this rank's sweep launch

    # ---- parity of what was just computed (frame 0, a sample of this rank's pixels)
    st = eng.stats()
    parity = None
    if rank == 0:
        from oracle import oracle_py

        got = d_power[0].cpu().numpy()
        pick = np.linspace(0, shard.pixel_count - 1, num=min(256, shard.pixel_count)).astype(np.int64)
        want = oracle_py.das_f32(host_frames[0], off[pick] + win_begin, frac[pick])
        floor = 1e-4 * want.max()
        parity = float((np.abs(got[pick] - want) / np.maximum(want, floor)).max())

    # ---- N > 1: the assembled heatmap of frame 0 must equal what the shards computed
    gather_check = None
    if world > 1:
        try:  # a failure here must not cost the run its result line
            full = sharding.gather_power(d_power[:1].contiguous(), sharding.all_shards(spec.res, spec.res, world), dst=0)
            if rank == 0:
                ok = full.shape == (1, spec.n_pixels) and torch.equal(full[0, : shard.pixel_count], d_power[0])
                gather_check = "ok" if ok else "MISMATCH: the assembled heatmap differs from rank 0's tile"
        except Exception as e:  # noqa: BLE001
            gather_check = f"failed: {type(e).__name__}: {e}"

    # ---- N > 1, second measurement: the frame-sharded decomposition (whole frames per rank, full grid;
    # each frame crosses xGMI once).  Reported beside the headline number, not instead of it.
    alt = None
    if world > 1 and B % world == 0 and os.environ.get("BENCH_ALT", "0") == "1":
        per = B // world
        first, _ = sharding.shard_frames(B, world, rank)
        off_all, frac_all = S.delay_table(spec, xyz, 0, spec.res)
        eng2 = pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, hist=hist, math=math, max_batch=per,
                          device=local_rank, grid_columns=spec.res)
        eng2.set_delay_table(off_all - win_begin, frac_all)
        eng2.set_active_mics(None)
        mine = tuple(torch.zeros((per, spec.n_mics, hist), dtype=torch.float32, device=dev) for _ in range(2))
        d_power2 = torch.zeros((per, spec.n_pixels), dtype=torch.float32, device=dev)
        scat = sharding.FrameScatterer(mine, bufs if rank == 0 else None, src=0)

        def post2(k):
            if rank == 0:
                bufs[k % 2].copy_(d_full[:, :, win_begin:win_begin + hist])
            scat.post(k)

        def run2(n):
            with torch.cuda.stream(stream):
                post2(0)
                for k in range(n):
                    frames = scat.wait(k)
                    if k + 1 < n:
                        post2(k + 1)
                    eng2.process_device(frames.data_ptr(), per, d_power2.data_ptr(), stream.cuda_stream)

        run2(W)
        fence()
        t0 = time.perf_counter()
        run2(K)
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # every rank's first frame, a sample of pixels, checked on rank 0 against the oracle
        pick = np.linspace(0, spec.n_pixels - 1, num=64).astype(np.int64)
        sample = d_power2[0, torch.from_numpy(pick).to(dev)].contiguous()
        got_all = [torch.empty_like(sample) for _ in range(world)] if rank == 0 else None
        dist.gather(sample, got_all, dst=0)
        if rank == 0:
            from oracle import oracle_py

            worst = 0.0
            for r in range(world):
                want = oracle_py.das_f32(host_frames[r * per], off_all[pick], frac_all[pick])
                worst = max(worst, float((np.abs(got_all[r].cpu().numpy() - want) / np.maximum(want, 1e-4 * want.max())).max()))
            alt = {"sharding": f"whole frames over {world} GPUs ({per} per rank per step, full grid each); rank 0 scatters "
                               f"{per * spec.n_mics * hist * 4 / 1e6:.1f} MB to each rank per step",
                   "value": B * K / float(t.item()), "unit": "frames/s", "ms_per_step": float(t.item()) / K * 1e3,
                   "parity_max_rel_err": worst}
        eng2.close()

    if rank == 0:
        fps = B * K / elapsed
        full_bytes = S.algorithmic_bytes_per_frame(spec.n_mics, spec.n_pixels, st.window)
        # dominant kernel = the sweep launch on this rank: algorithmic bytes of its slab x B frames
        launch_bytes = int(st.alg_bytes_frame) * B
        launch_flops = int(st.alg_flops_frame) * B
        ach_gbs = launch_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "heatmap frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": spec.name, "mics": spec.n_mics, "grid": f"{spec.res}x{spec.res}",
                "block_samples": 256, "frames_per_step": B, "math": args.math,
                "sharding": f"grid rows over {world} GPU(s); per step rank 0 broadcasts the {hist}-sample window "
                            f"of every mic ({B * spec.n_mics * hist * 4 / 1e6:.1f} MB), overlapped with the previous sweep"
                            if world > 1 else "single GPU",
                "alg_bytes_per_frame": full_bytes,
            },
            "roofline": {
                "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(spec.name, B, world),
                "kernel": "das sweep", "kernel_ms": kernel_ms, "launch_bytes": launch_bytes,
                "note": "the sweep is fp32-VALU/LDS bound (126 flop/B >> ridge), see valu",
            },
            "valu": {
                "achieved": launch_flops / (kernel_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": launch_flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
            },
            "parity_max_rel_err": parity,
        }
        if gather_check is not None:
            out["gather_check"] = gather_check
        if alt is not None:
            out["alt_sharding"] = alt
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: logic check only, not a scaling number"
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(S, spec, xyz, off, frac, host_frames[0], args.cpu_seconds)
            out["speedup_vs_cpu_1t"] = fps / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
