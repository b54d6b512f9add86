#!/usr/bin/env python3
"""bench.py -- heatmap frames/s of the delay-and-sum sweep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline] [--batch B]

One "step" = one pass of the hot path over one batch of B synthetic frames already resident
in HBM: (N > 1: RCCL broadcast of the batch from rank 0, overlapped with the previous
step's sweep) + one sweep launch per rank over that rank's slab of the steering grid.
The grid (total work) is fixed, so scaling is "strong".  Prints ONE JSON line on rank 0.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own
N ranks (one per GPU, `python -m torch.distributed.run` as a child process, before anything here has
touched the GPU) and exits with their status; started by torch.distributed.run itself (WORLD_SIZE
set) it is one of the ranks.

Workloads: c1, c2, headline, c3, c4 = BASELINE.json configs on the whole grid; c5 = configs[4], 1024 frames
in flight on ONE rank's slab of the 256x256 grid (1/8 of the rows; with N GPUs the N first slabs of 8N).

At N = 1 the line also carries, next to the batched `value`: "single_frame" (one frame per call, the
regime the reference's live display runs in), "pcie_inclusive" (awpu_hip_process on pageable host
buffers, upload and read-back inside the clock; never `value`), "bf16" (the bf16-accumulator mode on the
same frames: its error against the fp32 sweep and its rate) and "cpu_baseline" (the reference's own compiled
delay() on the host).  BENCH_ALT=1 (N > 1) adds a second, separately timed pass with whole frames per rank.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak fp32 vector
C5_SLABS = 8               # configs[4] shards the 256x256 grid over 8 GPUs


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", help="c1 | c2 | headline | c3 | c4 | c5")
    ap.add_argument("--batch", type=int, default=0, help="frames per step (per sweep launch); default 128, c5: 1024")
    ap.add_argument("--math", default="fast", choices=["fast", "exact", "bf16"])
    ap.add_argument("--interp", default="lerp", choices=["lerp", "fir8"], help="fir8: the 8-tap variant of delay()")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--no-extras", action="store_true", help="skip single_frame / pcie_inclusive / bf16")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--selftest-launcher", default="", help=argparse.SUPPRESS)  # tests: ok | fail
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """Start n ranks of this script under torch.distributed.run as a CHILD process (never an exec: this
    process may not replace itself once anything has touched the GPU, and nothing here has) and hand its
    output and exit status through.  Rank 0 prints the one JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between processes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), *argv]
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(mode: str) -> None:
    """What the launcher test runs in every rank instead of the bench: rendezvous over gloo, one
    collective, one line from rank 0 -- no GPU needed."""
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode == "fail" and rank == world - 1:
        sys.exit(3)
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "world": world, "sum": float(t.item())}), flush=True)
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ legs


def cpu_baseline(spec, off, frac, frame, seconds, interp="lerp"):
    """The reference's own delay() (oracle/_ref) in the loop nest of mimo.cpp:121-151, one
    thread, whole frames of the same workload until `seconds` have passed.  interp "fir8": the reference built
    without -mavx2, whose delay() is the 8-tap variant (delay.cpp:31-40) on its own filter.h table."""
    from oracle import oracle_py

    variant = "fir" if interp == "fir8" else "avx2"
    kind = "reference" if oracle_py.ref_available(variant) else "port"
    if kind == "reference":
        fps, frames = oracle_py.ref_bench(frame, off, frac, None, min_seconds=seconds, variant=variant)
    else:
        table = synthetic_fir_table()
        t0 = time.perf_counter()
        frames = 0
        while True:
            if interp == "fir8":
                oracle_py.das_fir8_f32(frame, off, frac, table)
            else:
                oracle_py.das_f32(frame, off, frac)
            frames += 1
            if time.perf_counter() - t0 >= seconds:
                break
        fps = frames / (time.perf_counter() - t0)
    flags = "-Ofast -ffast-math, no -mavx2: the 8-tap variant" if interp == "fir8" else "-Ofast -ffast-math -mavx2 -mfma"
    out = {
        "value": fps, "unit": "frames/s", "cores": 1, "kind": kind,
        "sample": f"{frames} whole frames of the same workload ({spec.name}), 1 thread, {os.cpu_count()} host cores "
                  f"present; the reference's delay.cpp compiled with its own flags ({flags}) "
                  f"minus -march=native, so that one .so runs on any host (its AVX2 intrinsics need no more)",
    }
    if kind == "reference":  # beside it: the same kernel with the pixels dealt to the box's CPU share
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()))
        fps_mt, frames_mt, _ = oracle_py.ref_bench_mt(frame, off, frac, threads, min_seconds=max(2.0, seconds / 4),
                                                      variant=variant)
        out["multi_thread"] = {"value": fps_mt, "unit": "frames/s", "cores": threads,
                               "sample": f"{frames_mt} whole frames, pixels dealt to {threads} threads (the reference's "
                                         f"MIMO worker itself is one thread)"}
    return out


def measured_traffic(workload, batch, world):
    """HBM-side bytes per launch from the committed PMC run of this round (profiles/r02_hbm_traffic.json,
    collected with tools/pmc_hbm.sh as MI355X_MICROARCH.md prescribes); (None, None) when it was not
    measured for this exact workload and batch."""
    for name in ("r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        path = REPO / "profiles" / name
        if world != 1 or not path.exists():
            continue
        rec = json.loads(path.read_text())
        if rec.get("workload") != workload:
            continue
        for m in rec["measurements"]:  # not proportional to the batch (L2 residency changes): exact matches only
            if m["frames_per_step"] == batch:
                return int(m["traffic_bytes_per_launch"]), f"profiles/{name} (PMC, tools/pmc_hbm.sh; not re-measured in this run)"
    return None, None


def synthetic_fir_table() -> np.ndarray:
    """A [101, 8] Blackman-windowed-sinc fractional-delay table (the kernels take the table as input: any
    table times the same)."""
    t = np.arange(8, dtype=np.float64)[None, :]
    d = (np.arange(101, dtype=np.float64) / 100.0)[:, None]
    x = t - 3.0 - d
    w = 0.42 - 0.5 * np.cos(2 * np.pi * (x + 4.0) / 8.0) + 0.08 * np.cos(4 * np.pi * (x + 4.0) / 8.0)
    h = np.sinc(x) * np.clip(w, 0.0, None)
    return (h / h.sum(axis=1, keepdims=True)).astype(np.float32)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # nothing above has touched the GPU
    if args.selftest_launcher:
        launcher_selftest(args.selftest_launcher)
        return
    # dmabuf IPC for RCCL between processes; read when the HSA runtime starts, so before torch touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    c5 = args.workload == "c5"
    spec = S.WORKLOADS["c4" if c5 else args.workload]
    B = args.batch or (1024 if c5 else 128)
    K, W = args.steps, args.warmup
    if c5 and args.steps == 20 and args.warmup == 3:
        K, W = 6, 2  # a 1024-frame step is 8x the default one
    math = {"fast": pkg.MATH_F32_FAST, "exact": pkg.MATH_F32_EXACT, "bf16": pkg.MATH_BF16_ACC}[args.math]
    interp = pkg.binding.INTERP_FIR8 if args.interp == "fir8" else pkg.binding.INTERP_LERP
    wl_name = (f"c5: 512 mics x 256x256 x 256, {B} frames in flight, one rank's slab of {C5_SLABS * world} "
               f"({spec.res // (C5_SLABS * world)} rows) per GPU") if c5 else spec.name

    # ---- one-off setup: geometry, this rank's slab of the delay table, frames in HBM
    if c5:
        shard = sharding.shard_rows(spec.res, spec.res, C5_SLABS * world, rank)
        grid_pixels = sum(sharding.shard_rows(spec.res, spec.res, C5_SLABS * world, r).pixel_count for r in range(world))
    else:
        shard = sharding.shard_rows(spec.res, spec.res, world, rank)
        grid_pixels = spec.n_pixels
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
        reach = 263 if args.interp == "fir8" else 257
        hist = ((int(hi.item()) - win_begin + reach + 3) // 4) * 4 + 4
        off = off - win_begin

    def make_engine(math_id, max_batch, off_, frac_, begin, count, grid_columns=spec.res):
        eng_ = pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, hist=hist, math=math_id, interp=interp,
                          max_batch=max_batch, device=local_rank, pixel_begin=begin, pixel_count=count,
                          grid_columns=0 if os.environ.get("BENCH_NO_GRID_HINT") else grid_columns)
        eng_.set_delay_table(off_, frac_)
        eng_.set_active_mics(None)
        if args.interp == "fir8":
            eng_.set_fir_table(synthetic_fir_table())
        return eng_

    eng = make_engine(math, B, off, frac, shard.pixel_begin, shard.pixel_count)

    # frames: generated 64 at a time (a 1024-frame batch of 512 mics is 2.1 GB), resident in HBM before the clock
    d_full = None
    host_first = None
    if rank == 0:
        d_full = torch.empty((B, spec.n_mics, pkg.binding.HIST), dtype=torch.float32, device=dev)  # the ingest layout
        for b0 in range(0, B, 64):
            n = min(64, B - b0)
            chunk = S.make_frames(xyz, n, seed=args.seed + b0)
            if b0 == 0:
                host_first = chunk[: min(n, 16)].copy()
            d_full[b0:b0 + n] = torch.from_numpy(chunk).to(dev)
        del chunk
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
    # How the batch gets to every rank, untimed: one broadcast per step, or a scatter of 1/N to every rank followed by
    # an all-gather (every xGMI link carries 1/N instead of the root's ring neighbour carrying all of it).  Which is
    # faster depends on the collective library's schedule for this topology, so both are timed over a few warm-up
    # steps and the faster one (the same on every rank: decided on the maximum over ranks) runs the timed region.
    # BENCH_BCAST=broadcast|scatter_allgather pins it.
    bcast_choice = {"mode": bcast.mode, "why": "BENCH_BCAST" if "BENCH_BCAST" in os.environ else "single mode"}
    trial_ranks = 2 if rehearsal else 4  # (two ranks have one link either way: nothing to choose)
    if world >= trial_ranks and "BENCH_BCAST" not in os.environ and B % world == 0:
        trial = {}
        for mode in ("broadcast", "scatter_allgather"):
            bcast = sharding.FrameBroadcaster(bufs, src=0, mode=mode)
            try:
                run_steps(1)
                fence()
                t1 = time.perf_counter()
                run_steps(max(2, W))
                fence()
                mine_s = time.perf_counter() - t1
            except (RuntimeError, ValueError, NotImplementedError) as exc:  # a backend without this collective:
                print(f"[bench] {mode} not usable here ({exc}); keeping the broadcast", file=sys.stderr)  # (every rank alike)
                trial[mode] = float("inf")
                continue
            tt = torch.tensor([mine_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[mode] = float(tt.item()) / max(2, W)
        best = min(trial, key=trial.get)
        bcast = sharding.FrameBroadcaster(bufs, src=0, mode=best)
        bcast_choice = {"mode": best, "why": "faster over the warm-up steps",
                        "ms_per_step": {m: (round(v * 1e3, 4) if v != float("inf") else None) for m, v in trial.items()}}
        run_steps(1)
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
    pick = np.linspace(0, shard.pixel_count - 1, num=min(256, shard.pixel_count)).astype(np.int64)
    if rank == 0:
        from oracle import oracle_py

        got = d_power[0].cpu().numpy()
        if args.interp == "fir8":
            want = oracle_py.das_fir8_f32(host_first[0], off[pick] + win_begin, frac[pick], synthetic_fir_table())
        elif args.math == "bf16":
            want = oracle_py.das_bf16acc(host_first[0], off[pick] + win_begin, frac[pick])
        else:
            want = oracle_py.das_f32(host_first[0], off[pick] + win_begin, frac[pick])
        floor = 1e-4 * want.max()
        parity = float((np.abs(got[pick] - want) / np.maximum(want, floor)).max())

    # ---- N > 1, second measurement: the frame-sharded decomposition (whole frames per rank, full grid;
    # each frame crosses xGMI once).  Reported beside the headline number, not instead of it.
    alt = None
    if world > 1 and B % world == 0 and not c5 and os.environ.get("BENCH_ALT", "0") == "1":
        per = B // world
        off_all, frac_all = S.delay_table(spec, xyz, 0, spec.res)
        eng2 = make_engine(math, per, off_all - win_begin, frac_all, 0, spec.n_pixels)
        mine = tuple(torch.zeros((per, spec.n_mics, hist), dtype=torch.float32, device=dev) for _ in range(2))
        d_power2 = torch.zeros((per, spec.n_pixels), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
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
        alt = {"sharding": f"whole frames over {world} GPUs ({per} per rank per step, full grid each); rank 0 scatters "
                           f"{per * spec.n_mics * hist * 4 / 1e6:.1f} MB to each rank per step",
               "value": B * K / float(t.item()), "unit": "frames/s", "ms_per_step": float(t.item()) / K * 1e3}
        eng2.close()

    out = None
    if rank == 0:
        fps = B * K / elapsed
        full_bytes = S.algorithmic_bytes_per_frame(spec.n_mics, grid_pixels, st.window)
        # dominant kernel = the sweep launch on this rank: algorithmic bytes of its slab x B frames
        launch_bytes = int(st.alg_bytes_frame) * B
        launch_flops = int(st.alg_flops_frame) * B
        if args.interp == "fir8":  # 16 flop per (pixel, mic, sample) instead of 4
            launch_flops = (16 * shard.pixel_count * st.usable * 256 + 6 * shard.pixel_count * 254) * B
        ach_gbs = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(spec.name, B, world)
        out = {
            "metric": "heatmap frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wl_name, "mics": spec.n_mics, "grid": f"{spec.res}x{spec.res}",
                "block_samples": 256, "frames_per_step": B, "math": args.math, "interp": args.interp,
                "sharding": f"grid rows over {world} GPU(s); per step rank 0 broadcasts the {hist}-sample window "
                            f"of every mic ({B * spec.n_mics * hist * 4 / 1e6:.1f} MB), overlapped with the previous sweep"
                            if world > 1 else "single GPU",
                "alg_bytes_per_frame": full_bytes,
            },
            "roofline": {
                "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "das sweep", "kernel_ms": kernel_ms, "launch_bytes": launch_bytes,
                "note": "HBM is the bound the metric names; the sweep itself is fp32-VALU bound (126 flop/B >> the "
                        "chip's ridge), its fraction of that peak is in \"valu\"",
            },
            "valu": {
                "achieved": launch_flops / (kernel_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": launch_flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                "note": "algorithmic flops (4 per pixel, mic and sample) over the measured kernel time",
            },
            "parity_max_rel_err": parity,
        }
        if alt is not None:
            out["alt_sharding"] = alt
        if world > 1:
            out["config"]["frame_exchange"] = bcast_choice
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: logic check only, not a scaling number"

    # ---- N = 1: the other regimes of the same path, and the CPU reference, beside the batched value
    if world == 1 and not args.no_extras and args.math == "fast" and args.interp == "lerp":
        from oracle import oracle_py

        n1 = 200 if shard.pixel_count * spec.n_mics <= (1 << 23) else 60
        eng1 = make_engine(math, 1, off, frac, shard.pixel_begin, shard.pixel_count)
        ev1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        d_p1 = torch.zeros((2, shard.pixel_count), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()  # (the fill runs on torch's stream; `stream` does not wait for it)
        with torch.cuda.stream(stream):
            for k in range(10):
                eng1.process_device(d_full[k % B].data_ptr(), 1, d_p1[1].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev1[0].record(stream)
            for k in range(n1):
                eng1.process_device(d_full[k % B].data_ptr(), 1, d_p1[1].data_ptr(), stream.cuda_stream)
            ev1[1].record(stream)
            torch.cuda.synchronize()
            wall1 = time.perf_counter() - t0
            eng1.process_device(d_full[0].data_ptr(), 1, d_p1[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        ms1 = ev1[0].elapsed_time(ev1[1]) / n1
        got1 = d_p1[0].cpu().numpy()
        want1 = oracle_py.das_f32(host_first[0], off[pick], frac[pick])
        out["single_frame"] = {
            "value": n1 / wall1, "unit": "frames/s", "calls": n1, "ms_per_frame_device": ms1,
            "valu_frac": int(st.alg_flops_frame) / (ms1 * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
            "hbm_frac": int(st.alg_bytes_frame) / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "parity_max_rel_err": float((np.abs(got1[pick] - want1) / np.maximum(want1, 1e-4 * want1.max())).max()),
            "note": "one frame per call (asynchronous device-pointer entry, back to back): the regime of the reference's live path",
        }
        eng1.close()

        # host-buffer entry: pageable host memory in, power out, upload and read-back inside the clock
        nb = min(B, 128)
        host_batch = d_full[:nb].cpu().numpy()
        eng.process(host_batch)  # allocates the staging buffers
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.process(host_batch)
        wallp = (time.perf_counter() - t0) / reps
        out["pcie_inclusive"] = {"value": nb / wallp, "unit": "frames/s", "frames_per_call": nb, "ms_per_call": wallp * 1e3,
                                 "note": "awpu_hip_process on pageable host buffers (only the touched window of every "
                                         "stream is uploaded); never the headline value"}

        # bf16 accumulator on the same frames: error against the fp32 sweep just run, and its own rate
        nf = min(B, 16)
        eng16 = make_engine(pkg.MATH_BF16_ACC, nf, off, frac, shard.pixel_begin, shard.pixel_count)
        d_p16 = torch.zeros((nf, shard.pixel_count), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ev16 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        with torch.cuda.stream(stream):
            eng16.process_device(d_full.data_ptr(), nf, d_p16.data_ptr(), stream.cuda_stream)
            ev16[0].record(stream)
            for _ in range(3):
                eng16.process_device(d_full.data_ptr(), nf, d_p16.data_ptr(), stream.cuda_stream)
            ev16[1].record(stream)
            torch.cuda.synchronize()
        ms16 = ev16[0].elapsed_time(ev16[1]) / 3
        p32 = d_power[:nf].cpu().numpy().astype(np.float64)
        p16 = d_p16.cpu().numpy().astype(np.float64)
        floor = 1e-4 * p32.max(axis=1, keepdims=True)
        out["bf16"] = {
            "max_rel_err": float((np.abs(p16 - p32) / np.maximum(p32, floor)).max()),
            "max_err_over_frame_peak": float((np.abs(p16 - p32) / p32.max(axis=1, keepdims=True)).max()),
            "frames_s": nf / (ms16 * 1e-3), "frames": nf, "fp32_frames_s_same_run": B / (kernel_ms * 1e-3),
            "note": "AWPU_MATH_BF16_ACC: running sums kept in bf16 (round to nearest even after every mic), everything else "
                    "fp32; max_rel_err is per pixel against the fp32 sweep of the same frames with the metric of the parity "
                    "tests (relative to max(pixel, 1e-4 x frame peak): beam nulls dominate it), no gate; gfx950 has no packed bf16 add, "
                    "so the mode runs in the exact-order kernel's structure",
        }
        eng16.close()

    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(spec, off, frac, host_first[0], args.cpu_seconds, args.interp)
        out["cpu_baseline"]["pixels"] = int(shard.pixel_count)
        out["speedup_vs_cpu_1t"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)

    # ---- N > 1, opt-in: the assembled heatmap of frame 0 must equal what the shards computed (a collective
    # after the result line, so that a rank that fails here cannot cost the run its line)
    if world > 1 and os.environ.get("BENCH_GATHER_CHECK") == "1":
        shards = [sharding.shard_rows(spec.res, spec.res, C5_SLABS * world if c5 else world, r) for r in range(world)]
        full = sharding.gather_power(d_power[:1].contiguous(), shards, dst=0)
        if rank == 0:
            ok = full.shape == (1, grid_pixels) and torch.equal(full[0, : shard.pixel_count], d_power[0])
            print("gather check:", "ok" if ok else "MISMATCH: the assembled heatmap differs from rank 0's tile", file=sys.stderr)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
