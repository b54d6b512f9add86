#!/usr/bin/env python3
"""bench.py -- heatmap frames/s of the delay-and-sum sweep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline] [--batch B]

One "step" = one pass of the hot path over one batch of B synthetic frames already resident in HBM:
(N > 1: the exchange of the batch from rank 0 -- RCCL over xGMI, overlapped with the previous step's
sweep --) + one sweep launch per rank over that rank's slab of the steering grid.  Prints ONE JSON
line on rank 0.

Work per step.  The grid's rows are split over the N ranks and the batch grows with N (B = 128 N unless
--batch says otherwise), so that every rank's launch keeps the size it has on one GPU (128 frames x the
whole grid = 1024 frames x an eighth of it: ~4.8 ms at the headline) and launch / collective latency is
amortised alike at every N: per-GPU work is fixed, `"scaling": "weak"`; `value` is whole frames (whole
P-pixel heatmaps) per second of the job.  With --batch given the batch is fixed and the scaling "strong".

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own
N ranks (one per GPU, `python -m torch.distributed.run` as a child process, before anything here has
touched the GPU) and exits with their status; started by torch.distributed.run itself (WORLD_SIZE
set) it is one of the ranks.  It refuses to start children under a profiler (rocprofv3's preloaded
library has initialised the GPU before main() runs): profile one rank's slab instead (--workload c5).

Workloads: c1, c2, headline, c3, c4 = BASELINE.json configs on the whole grid; c5 = configs[4], 1024 frames
in flight on ONE rank's slab of the 256x256 grid (1/8 of the rows; with N GPUs the N first slabs of 8N).

At N = 1 the line also carries, next to the batched `value`:
  "parity"            GPU vs the oracle on EVERY pixel of frame 0, unfloored (tests/util.parity_report); "dc_ok": the run's math mode
                      within 1e-5 of the reference-built goldens at every DC offset (parity_dc)
  "single_frame"      one frame per call, the regime the reference's live display runs in; "other_mode": the same in the other fp32 mode
  "pcie_inclusive"    awpu_hip_process on pageable host buffers, upload and read-back inside the clock; never `value`
  "bf16"              the bf16-accumulator mode on the same frames: its ERROR against the fp32 sweep
  "parity_dc"         both math modes on the reference-generated DC-biased goldens: error per offset (exact <= 1e-5 flat)
  "single_frame_c2"   BASELINE configs[1]'s shape (256 mics, 64x64) one frame per call, in the run's mode
  "reference_default" the shape the reference ships (64 mics, 100x100, one frame per call) with its CPU time and the
                      5.24 ms real-time budget of a block beside it, in both fp32 modes
  "workloads"         the headline shape in the OTHER fp32 mode (--math fast when the run is the default), then c2, c3 and the c5
                      slab in the run's mode and c3 with the 8-tap FIR variant, a few steps each: kernel ms, VALU fraction, full-grid parity
  "projected_scaling" rank 0's slab of an 8-rank run through the N > 1 step loop, the collective replaced by a local
                      copy of the same bytes: what one GPU can say about the 8-GPU step (NOT a scaling measurement)
  "projected_scaling_strong"  the same for configs[3]: 512 mics x 256x256 at a FIXED batch, an eighth of the grid per GPU
  "cpu_baseline"      the reference's own compiled delay() on the host
BENCH_ALT=1 (N > 1) adds a second, separately timed pass with whole frames per rank.
"""
from __future__ import annotations

import argparse
import datetime
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
for _p in (str(REPO), str(REPO / "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak fp32 vector
C5_SLABS = 8               # configs[4] shards the 256x256 grid over 8 GPUs
FRAMES_PER_RANK_STEP = 128  # default batch = this x world: a rank's launch keeps its one-GPU size
DISTINCT_FRAMES = 128      # synthetic frames generated; larger batches repeat them (every copy is swept in full)
RENDEZVOUS_TIMEOUT_S = 180
RUN_TIMEOUT_S = 900        # N > 1: everything after the rendezvous, per rank


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", help="c1 | c2 | headline | c3 | c4 | c5")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step; default 128 x GPUs (c5: 1024 x GPUs), which keeps every rank's launch at its one-GPU size")
    ap.add_argument("--math", default="exact", choices=["exact", "fast", "bf16"],
                    help="exact (the library's default): the reference's operation and mic order, within 1e-5 of it on any input; "
                         "fast: the re-ordered fp32 sweep (opt-in: within 1e-5 on zero-mean input only)")
    ap.add_argument("--interp", default="lerp", choices=["lerp", "fir8"], help="fir8: the 8-tap variant of delay()")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip single_frame / pcie_inclusive / bf16 / workloads / projected_scaling")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--selftest-launcher", default="", help=argparse.SUPPRESS)  # tests: ok | fail
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_profiler(env=None) -> str:
    """Non-empty (the reason) when this process runs under rocprofv3 or one of the repo's profiling scripts: the
    profiler's preloaded library has initialised the GPU before main() runs, and a process in that state must not
    start another program (this pool forbids it; the box can go down)."""
    env = os.environ if env is None else env
    if env.get("AWPU_UNDER_PROFILER") == "1":  # (NOT AWPU_NO_BUILD: that one only stops rebuilds and says nothing about a profiler)
        return "AWPU_UNDER_PROFILER=1 (set by tools/pmc.sh, pmc_hbm.sh, gpu_profile.sh: a profiler is in the picture)"
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        return "LD_PRELOAD carries the rocprofiler tool library"
    for key in env:
        if key.startswith(("ROCPROF", "ROCP_", "ROCPROFILER", "ROCTRACER")):
            return f"{key} is set (rocprofv3 environment)"
    return ""


def launch_ranks(n: int, argv) -> int:
    """Start n ranks of this script under torch.distributed.run as a CHILD process (never an exec: this
    process may not replace itself once anything has touched the GPU, and nothing here has) and hand its
    output and exit status through.  Rank 0 prints the one JSON line."""
    why = under_profiler()
    if why:
        print(f"bench.py --gpus {n}: refusing to start child ranks under a profiler ({why}).  Profile one rank's slab "
              f"instead: `--workload c5` at --gpus 1, or start the ranks with torch.distributed.run yourself.", file=sys.stderr)
        return 2
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


class Watchdog:
    """Exit non-zero when a rendezvous or a first collective does not come back: a rank that waits for a peer that
    never arrives would otherwise sit there until the driver's own limit."""

    def __init__(self, seconds: float, what: str, status: int = 4):
        self.timer = threading.Timer(seconds, self._fire, args=(seconds, what, status))
        self.timer.daemon = True

    @staticmethod
    def _fire(seconds, what, status):
        print(f"[bench] {what} did not finish within {seconds:.0f} s: giving up (exit {status})", file=sys.stderr, flush=True)
        os._exit(status)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()


def default_batch(world: int, c5: bool) -> int:
    """Frames per step when --batch is not given: the one-GPU batch x the number of ranks, so that a rank's launch
    (B frames x 1/world of the grid) keeps the size -- and the launch and collective overheads the share -- it has
    at N = 1."""
    return (1024 if c5 else FRAMES_PER_RANK_STEP) * world


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


def measured_traffic(workload, batch, world, math="fast"):
    """HBM-side bytes per launch from the committed PMC run (profiles/r0N_hbm_traffic.json, collected with
    tools/pmc_hbm.sh as MI355X_MICROARCH.md prescribes; newest round first); (None, None) when it was not
    measured for this exact workload, batch and math mode."""
    for name in ("r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        path = REPO / "profiles" / name
        if world != 1 or not path.exists():
            continue
        rec = json.loads(path.read_text())
        if rec.get("workload") != workload:
            continue
        for m in rec["measurements"]:  # not proportional to the batch (L2 residency changes): exact matches only
            if m["frames_per_step"] == batch and m.get("math", "fast") == math:
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


def full_grid_parity(got, frame, off, frac, math="fast", interp="lerp"):
    """GPU powers of one frame against the oracle on EVERY pixel handed in, unfloored: tests/util.parity_report."""
    import util
    from oracle import oracle_py

    if interp == "fir8":
        table = synthetic_fir_table()
        r32, r64 = oracle_py.das_fir8_f32(frame, off, frac, table), oracle_py.das_fir8_f64(frame, off, frac, table)
    elif math == "bf16":  # the mode's own checker (same operations, same order); no exact sums to compare with
        r32, r64 = oracle_py.das_bf16acc(frame, off, frac), None
    else:
        r32, r64 = oracle_py.das_f32(frame, off, frac), oracle_py.das_f64(frame, off, frac)
    return util.parity_report(got, r32, r64)


class RankJob:
    """Everything one rank of a `world`-rank run owns: its share of the grid's rows and of the delay table, its engine,
    the double-buffered frame exchange and the step loop.  world = 1 is the single-GPU bench.  `stub` replaces the
    collective by a local copy of the same bytes (projected_scaling: the N > 1 loop on one GPU).

    Rows.  N > 1: row groups of four dealt round-robin (sharding.shard_rows_interleaved: edge rows cost more than
    centre rows, an N-GPU step takes as long as its slowest rank); BENCH_SHARDS=contiguous gives plain slabs.
    Exchange.  "packed" (default where the library offers it): rank 0 runs the sweep's pack pass once per batch, on a
    side stream beside the previous sweep, the packed frame pairs travel, and every rank sweeps them as they arrive
    (awpu_hip_pack_frames / awpu_hip_process_packed) -- no window cut on the root, no pack pass on the others.
    "window" (BENCH_EXCHANGE=window, FIR8, exact math): rank 0 cuts the touched window of every mic out of its
    snapshots, that travels, every rank runs the whole sweep (pack included) on it."""

    def __init__(self, pkg, sharding, torch, dist, args, spec, world, rank, dev, local_rank, B, c5=False, stub=False,
                 frames_src=None):
        self.pkg, self.sharding, self.torch, self.dist = pkg, sharding, torch, dist
        self.args, self.spec, self.world, self.rank, self.dev, self.B, self.stub = args, spec, world, rank, dev, B, stub
        S = pkg.synthetic
        slabs = C5_SLABS * world if c5 else world
        interleave = world > 1 and not c5 and os.environ.get("BENCH_SHARDS", "interleaved") != "contiguous"
        self.shards = sharding.all_shards(spec.res, spec.res, slabs, interleaved=interleave)[:world]
        self.shard = self.shards[rank]
        self.grid_pixels = sum(s.pixel_count for s in self.shards) if c5 else spec.n_pixels
        self.xyz = S.geometry(spec)
        off, frac = S.delay_table_for(spec, self.xyz, self.shard.row_ranges)
        self.off_abs = off
        self.hist = pkg.binding.HIST
        self.win_begin, self.win_hist, self.window = 0, pkg.binding.HIST, None
        self.exchange = None
        if world > 1:
            # Only the window [min off, max off + 257) of every mic is ever read, so that is all that
            # travels, in either exchange format; every rank needs the union over the slabs.
            lo, hi = int(off.min()), int(off.max())
            if stub:  # what the all-reduce below would find (a few table rows across the grid suffice)
                for r in range(0, spec.res, max(1, spec.res // 16)):
                    o, _ = S.delay_table(spec, self.xyz, r, 1)
                    lo, hi = min(lo, int(o.min())), max(hi, int(o.max()))
            else:
                t_lo = torch.tensor([lo], dtype=torch.int64, device=dev)
                t_hi = torch.tensor([hi], dtype=torch.int64, device=dev)
                dist.all_reduce(t_lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(t_hi, op=dist.ReduceOp.MAX)
                lo, hi = int(t_lo.item()), int(t_hi.item())
            reach = 263 if args.interp == "fir8" else 257
            self.win_begin = lo
            self.win_hist = ((hi - lo + reach + 3) // 4) * 4 + 4
            self.window = (lo, hi + reach)
            self.exchange = os.environ.get("BENCH_EXCHANGE", "packed")
            if args.math == "bf16" or args.interp != "lerp" or B % 2:  # (both fp32 modes have a packed form: awpu_hip_pack_frames)
                self.exchange = "window"
        self.math = {"fast": pkg.MATH_F32_FAST, "exact": pkg.MATH_F32_EXACT, "bf16": pkg.MATH_BF16_ACC}[args.math]
        self.interp = pkg.binding.INTERP_FIR8 if args.interp == "fir8" else pkg.binding.INTERP_LERP
        self.local_rank = local_rank
        self.eng = None
        self.root_work = True  # (projected_scaling switches it off to time what a rank other than the ingest rank does)
        if self.exchange == "packed":  # snapshots stay in the ingest layout on rank 0; every rank stages the union window
            self.eng = self.make_engine(self.math, B, off, frac, self.shard.pixel_begin, self.shard.pixel_count, window=self.window)
            try:
                packed_floats = self.eng.packed_bytes(B) // 4
            except pkg.AwpuError as exc:  # this table / mic list has no packed form: exchange raw windows
                if exc.status != pkg.binding.ERR_STATE:
                    raise
                self.eng.close()
                self.eng, self.exchange = None, "window"
        if self.eng is None:
            if self.exchange == "window":  # every rank sweeps [B][mics][Wc] windows: hist = Wc, offsets relative to the window
                self.hist = self.win_hist
                off = off - self.win_begin
            self.eng = self.make_engine(self.math, B, off, frac, self.shard.pixel_begin, self.shard.pixel_count)
        self.off, self.frac = off, frac

        # frames: DISTINCT_FRAMES synthetic ones generated 64 at a time, repeated up to B, resident in HBM before the clock
        self.d_full = frames_src
        self.host_first = None
        if rank == 0 and self.d_full is None:
            self.d_full = torch.empty((B, spec.n_mics, pkg.binding.HIST), dtype=torch.float32, device=dev)  # the ingest layout
            distinct = min(B, DISTINCT_FRAMES)
            for b0 in range(0, distinct, 64):
                n = min(64, distinct - b0)
                chunk = S.make_frames(self.xyz, n, seed=args.seed + b0)
                if b0 == 0:
                    self.host_first = chunk[: min(n, 16)].copy()
                self.d_full[b0:b0 + n] = torch.from_numpy(chunk).to(dev)
            for b0 in range(distinct, B, distinct):
                n = min(distinct, B - b0)
                self.d_full[b0:b0 + n] = self.d_full[:n]
        # a real (non-null) stream: its handle goes to the C ABI, and the torch events that time the sweep are
        # recorded on the same stream; `aux` carries the root's pack / window cut and the collectives' enqueue point.
        # Both at the default priority: measured through projected_scaling (round 3, two alternating runs on one
        # box), step wall of the root / of a peer: sweep above aux 4.99-5.03 / 4.68-4.69 ms, aux above sweep 4.99-5.00 /
        # 4.55-4.59, equal 4.93-4.95 / 4.55-4.56 -- a starved side stream delivers the next batch late, a starved sweep
        # is simply slower.  BENCH_STREAM_PRIO="sweep,aux" overrides for tuning runs.
        prio = [int(x) for x in os.environ.get("BENCH_STREAM_PRIO", "0,0").split(",")]
        self.stream = torch.cuda.Stream(device=dev, priority=prio[0])
        self.aux = torch.cuda.Stream(device=dev, priority=prio[1])
        self.swept = [torch.cuda.Event(), torch.cuda.Event()]
        if self.exchange == "packed":
            pairs = B // 2
            self.bufs = tuple(torch.zeros((pairs, packed_floats // pairs), dtype=torch.float32, device=dev) for _ in range(2))
            self.exchange_mb = packed_floats * 4 / 1e6
        elif self.exchange == "window":
            self.bufs = tuple(torch.zeros((B, spec.n_mics, self.hist), dtype=torch.float32, device=dev) for _ in range(2))
            self.exchange_mb = B * spec.n_mics * self.hist * 4 / 1e6
        else:
            self.bufs = (self.d_full, self.d_full)
            self.exchange_mb = 0.0
        self.d_power = torch.zeros((B, self.shard.pixel_count), dtype=torch.float32, device=dev)
        # "raw_scatter" (sharding.RawScatterExchange): the root's pack pass spread over the ranks -- every rank but the root
        # receives batch / world raw snapshots per step and packs them itself
        self.raw = None
        self.raw_ok = self.exchange == "packed" and world > 1 and (B // 2) % world == 0
        if self.raw_ok and (rank != 0 or stub == "raw_scatter"):
            self.raw = tuple(torch.empty((B // world, spec.n_mics, pkg.binding.HIST), dtype=torch.float32, device=dev) for _ in range(2))
        if stub:
            # stands for the bytes the collective would deliver (the real frames, so that the sweep sees real samples)
            self.arrival = torch.empty_like(self.bufs[0])
            torch.cuda.synchronize()
            self.fill(self.arrival, torch.cuda.current_stream(dev))
            raw_stub = None
            if stub == "raw_scatter":  # this rank's slice arrives raw and is packed here; the other slots arrive packed
                raw_stub = (self.raw[0], self.d_full[:B // world], self.pack_slice, 0, world)
            self.bcast = sharding.LocalCopyExchange(self.bufs, self.arrival, priority=prio[1], raw_scatter=raw_stub)
        else:
            self.bcast = self.make_exchange(os.environ.get("BENCH_BCAST", "broadcast"))
        torch.cuda.synchronize()

    def pack_slice(self, raw, slot):
        """awpu_hip_pack_frames for a slice of the batch, on the calling stream: raw snapshots [n, mics, 1024] -> their n / 2 packed pairs"""
        self.eng.pack_frames(raw.data_ptr(), raw.shape[0], slot.data_ptr(), self.torch.cuda.current_stream(self.dev).cuda_stream)

    def make_exchange(self, mode):
        if mode == "raw_scatter":
            if not self.raw_ok:
                raise ValueError("raw_scatter needs the packed exchange format and whole frame pairs per rank")
            return self.sharding.RawScatterExchange(self.bufs, self.raw, self.d_full if self.rank == 0 else None, self.pack_slice, src=0)
        return self.sharding.FrameBroadcaster(self.bufs, src=0, mode=mode)

    def make_engine(self, math_id, max_batch, off_, frac_, begin, count, grid_columns=None, window=None, hist=None):
        spec = self.spec
        grid_columns = spec.res if grid_columns is None else grid_columns
        eng_ = self.pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, hist=hist or self.hist, math=math_id,
                               interp=self.interp, max_batch=max_batch, device=self.local_rank, pixel_begin=begin,
                               pixel_count=count, window=window,
                               grid_columns=0 if os.environ.get("BENCH_NO_GRID_HINT") else grid_columns)
        eng_.set_delay_table(off_, frac_)
        eng_.set_active_mics(None)
        if self.args.interp == "fir8":
            eng_.set_fir_table(synthetic_fir_table())
        return eng_

    def fill(self, buf, on):
        """The root's share of a step: the resident snapshots -> the exchange buffer, enqueued on stream `on`."""
        if self.exchange == "packed":
            self.eng.pack_frames(self.d_full.data_ptr(), self.B, buf.data_ptr(), on.cuda_stream)
        else:
            with self.torch.cuda.stream(on):
                buf.copy_(self.d_full[:, :, self.win_begin:self.win_begin + self.hist])

    def post(self, k):
        """Start batch k on its way into buffer k % 2.  Everything here runs on the side stream: it waits for the sweep
        that last read the buffer (step k - 2), not for the sweep in progress, so the root's pack pass and the
        collective run beside sweep k - 1."""
        if self.exchange is None:
            return
        if k >= 2:
            self.aux.wait_event(self.swept[k % 2])
        if self.rank == 0 and self.root_work and "raw_scatter" not in self.bcast.mode:
            self.fill(self.bufs[k % 2], self.aux)
        with self.torch.cuda.stream(self.aux):  # (a collective is ordered after the work of the stream it is called on)
            self.bcast.post(k)

    def sweep(self, frames):
        if self.exchange == "packed":
            self.eng.process_packed(frames.data_ptr(), self.B, self.d_power.data_ptr(), self.stream.cuda_stream)
        else:
            self.eng.process_device(frames.data_ptr(), self.B, self.d_power.data_ptr(), self.stream.cuda_stream)

    def run_steps(self, n, ev=None):
        self.aux.wait_stream(self.stream)  # (sweeps of an earlier call may still read the buffers)
        with self.torch.cuda.stream(self.stream):
            self.post(0)
            for k in range(n):
                frames = self.bcast.wait(k)
                if k + 1 < n:
                    self.post(k + 1)  # next batch travels while this one is swept
                if ev is not None:
                    ev[0][k].record(self.stream)
                self.sweep(frames)
                if ev is not None:
                    ev[1][k].record(self.stream)
                self.swept[k % 2].record(self.stream)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1 and not self.stub:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, K, W):
        """W untimed steps, then exactly K timed ones: (wall seconds, mean sweep-launch ms by events on the launch stream)."""
        torch = self.torch
        self.run_steps(W)
        self.fence()
        events = ([torch.cuda.Event(enable_timing=True) for _ in range(K)],
                  [torch.cuda.Event(enable_timing=True) for _ in range(K)])
        t0 = time.perf_counter()
        self.run_steps(K, events)
        t_enqueued = time.perf_counter() - t0  # the host is done: what is left is the device catching up
        self.fence()
        elapsed = time.perf_counter() - t0
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(*events)]))
        return elapsed, kernel_ms, t_enqueued

    def close(self):
        self.eng.close()


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
        # a peer that never arrives, or a first collective that hangs, ends the run with a non-zero status
        with Watchdog(RENDEZVOUS_TIMEOUT_S + 60, "rendezvous + first collective"):
            timeout = datetime.timedelta(seconds=RENDEZVOUS_TIMEOUT_S)
            if rehearsal:
                dist.init_process_group("gloo", timeout=timeout)
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=timeout)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise SystemExit(f"first all-reduce returned {probe.item()} on rank {rank}, expected {world}")

    pkg = importlib.import_module("beamforming-lk_amd")
    sharding = importlib.import_module("beamforming-lk_amd.sharding")
    S = pkg.synthetic
    c5 = args.workload == "c5"
    spec = S.WORKLOADS["c4" if c5 else args.workload]
    B = args.batch or default_batch(world, c5)
    K, W = args.steps, args.warmup
    if c5 and args.steps == 20 and args.warmup == 3:
        K, W = 6, 2  # a 1024-frame step is 8x the default one
    wl_name = (f"c5: 512 mics x 256x256 x 256, {B} frames in flight, one rank's slab of {C5_SLABS * world} "
               f"({spec.res // (C5_SLABS * world)} rows) per GPU") if c5 else spec.name

    # N > 1: a collective that never completes (a rank that died, a link that hangs) ends this rank with status 4 instead
    # of sitting in the driver's own limit; generous -- table builds, K steps and the exchange trial fit many times over
    run_guard = Watchdog(RUN_TIMEOUT_S, f"the {world}-rank run") if world > 1 else None
    if run_guard:
        run_guard.__enter__()
    job = RankJob(pkg, sharding, torch, dist, args, spec, world, rank, dev, local_rank, B, c5=c5)
    shard, off, frac, eng = job.shard, job.off, job.frac, job.eng
    d_full, d_power, bufs, stream, host_first = job.d_full, job.d_power, job.bufs, job.stream, job.host_first
    win_begin, win_hist, grid_pixels, xyz = job.win_begin, job.win_hist, job.grid_pixels, job.xyz
    run_steps, fence = job.run_steps, job.fence

    run_steps(W)
    fence()
    # How the batch gets to every rank, untimed: one broadcast per step, or a scatter of 1/N to every rank followed by
    # an all-gather (every xGMI link carries 1/N instead of the root's ring neighbour carrying all of it).  Which is
    # faster depends on the collective library's schedule for this topology, so both are timed over a few warm-up
    # steps and the faster one (the same on every rank: decided on the maximum over ranks) runs the timed region.
    # A third schedule, raw_scatter (sharding.RawScatterExchange), spreads the root's pack pass over the ranks at the price of
    # raw snapshots on the wire.  BENCH_BCAST=broadcast|scatter_allgather|raw_scatter pins one.
    bcast_choice = {"mode": job.bcast.mode, "why": "BENCH_BCAST" if "BENCH_BCAST" in os.environ else "single mode"}
    trial_ranks = 2 if rehearsal else 4  # (two ranks have one link either way: nothing to choose)
    if world >= trial_ranks and "BENCH_BCAST" not in os.environ and B % world == 0:
        trial = {}
        for mode in ("broadcast", "scatter_allgather") + (("raw_scatter",) if job.raw_ok else ()):
            try:
                job.bcast = job.make_exchange(mode)
            except Exception as exc:  # noqa: BLE001 -- (the same on every rank: a property of the configuration)
                print(f"[bench] {mode} not available on rank {rank} ({exc})", file=sys.stderr)
                continue
            mine_s = float("inf")
            try:
                run_steps(1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run_steps(max(2, W))
                torch.cuda.synchronize()
                mine_s = time.perf_counter() - t1
            except Exception as exc:  # noqa: BLE001 -- a backend without this collective, or any other failure of an OPTIONAL schedule
                print(f"[bench] {mode} not usable on rank {rank} ({exc}); keeping the broadcast", file=sys.stderr)
            # EVERY rank reaches this all-reduce, whichever way its trial ended: one that failed votes "infinitely slow",
            # so that all ranks agree on the mode and none waits in a collective the others skipped
            tt = torch.tensor([mine_s], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[mode] = float(tt.item()) / max(2, W)
        best = min(trial, key=trial.get) if min(trial.values()) != float("inf") else "broadcast"
        job.bcast = job.make_exchange(best)
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
    step_kernel_ms = [a.elapsed_time(b) for a, b in zip(*events)]  # this rank's sweep launch, every timed step
    kernel_ms = float(np.mean(step_kernel_ms))

    # ---- parity of what was just computed: frame 0, EVERY pixel of this rank's slab, unfloored
    st = eng.stats()
    parity = None
    if rank == 0:
        parity = full_grid_parity(d_power[0].cpu().numpy(), host_first[0], job.off_abs, frac, args.math, args.interp)
        parity["what"] = (f"frame 0 of the timed batch, all {shard.pixel_count} pixels of rank 0's slab, GPU vs oracle.das_f32 "
                          f"(the reference's operations) and vs exact fp64 sums; `ok` = max_rel_unfloored <= 1e-5, flat")

    # ---- N > 1, second measurement: the frame-sharded decomposition (whole frames per rank, full grid;
    # each frame crosses xGMI once).  Reported beside the headline number, not instead of it.
    alt = None
    if world > 1 and B % world == 0 and not c5 and os.environ.get("BENCH_ALT", "0") == "1":
        per = B // world
        off_all, frac_all = S.delay_table(spec, xyz, 0, spec.res)
        eng2 = job.make_engine(job.math, per, off_all - win_begin, frac_all, 0, spec.n_pixels, hist=win_hist)
        mine = tuple(torch.zeros((per, spec.n_mics, win_hist), dtype=torch.float32, device=dev) for _ in range(2))
        full2 = tuple(torch.zeros((B, spec.n_mics, win_hist), dtype=torch.float32, device=dev) for _ in range(2)) if rank == 0 else None
        d_power2 = torch.zeros((per, spec.n_pixels), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        scat = sharding.FrameScatterer(mine, full2, src=0)

        def post2(k):
            if rank == 0:
                full2[k % 2].copy_(d_full[:, :, win_begin:win_begin + win_hist])
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
                           f"{per * spec.n_mics * win_hist * 4 / 1e6:.1f} MB to each rank per step",
               "value": B * K / float(t.item()), "unit": "frames/s", "ms_per_step": float(t.item()) / K * 1e3}
        eng2.close()
        del full2, mine

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
        traffic, traffic_src = measured_traffic(spec.name, B, world, args.math) if args.interp == "lerp" else (None, None)
        out = {
            "metric": "heatmap frames/sec", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": wl_name, "mics": spec.n_mics, "grid": f"{spec.res}x{spec.res}",
                "block_samples": 256, "frames_per_step": B, "math": args.math, "interp": args.interp,
                "math_note": {"exact": "AWPU_MATH_F32_EXACT, the library's default: delay.cpp:19-25's operations in its order, mics in "
                                       "antenna.index[] order -- within 1e-5 of the reference on any input",
                              "fast": "AWPU_MATH_F32_FAST, an explicit opt-in: re-ordered fp32 (stencil first); within 1e-5 of the reference "
                                      "on zero-mean input only (parity_dc)",
                              "bf16": "AWPU_MATH_BF16_ACC: BASELINE configs[4]'s accumulator experiment, not for production"}[args.math],
                "frames_per_step_rule": ("--batch given: fixed batch" if args.batch else
                                         f"{B // world} x {world} GPU(s): the batch grows with the number of ranks so that a rank's "
                                         f"launch ({B} frames x 1/{world} of the grid) keeps its one-GPU size"),
                "distinct_frames": min(B, DISTINCT_FRAMES),
                "sharding": (f"grid rows over {world} GPUs, " +
                             ("row groups of four dealt round-robin (edge and centre rows alike on every rank)" if shard.ranges
                              else "contiguous slabs") +
                             f"; per step rank 0 sends the touched window of every mic ({job.exchange_mb:.1f} MB, " +
                             ("as packed frame pairs: packed once on rank 0, swept as they arrive" if job.exchange == "packed"
                              else f"{win_hist} samples per mic, cut out of the snapshots") +
                             ") to every rank, overlapped with the previous sweep") if world > 1 else "single GPU",
                "alg_bytes_per_frame": full_bytes,
            },
            "roofline": {
                "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "das sweep: " + pkg.binding.KERNEL_NAMES[st.kernel_variant], "kernel_ms": kernel_ms, "kernel_ms_min": float(np.min(step_kernel_ms)),
                "kernel_ms_median": float(np.median(step_kernel_ms)), "launch_bytes": launch_bytes,
                "note": "HBM is the bound the metric names; the sweep itself is fp32-VALU bound (126 flop/B >> the "
                        "chip's ridge), its fraction of that peak is in \"valu\"",
            },
            "valu": {
                "achieved": launch_flops / (kernel_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": launch_flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                "note": "algorithmic flops (4 per pixel, mic and sample) over the measured kernel time",
            },
            "parity_max_rel_err": parity["max_rel_unfloored"],
            "parity": parity,
        }
        if alt is not None:
            out["alt_sharding"] = alt
        if world > 1:
            out["config"]["frame_exchange"] = bcast_choice
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: logic check only, not a scaling number"

    # ---- N = 1: the other regimes of the same path, and the CPU reference, beside the batched value
    if world == 1 and not args.no_extras and args.math in ("exact", "fast") and args.interp == "lerp":
        other = "fast" if args.math == "exact" else "exact"
        math_ids = {"fast": pkg.MATH_F32_FAST, "exact": pkg.MATH_F32_EXACT}

        def single_frame_leg(mode):
            """one frame per call, back to back on the launch stream, in fp32 mode `mode`"""
            n1 = 200 if shard.pixel_count * spec.n_mics <= (1 << 23) else 60
            eng1 = job.make_engine(math_ids[mode], 1, off, frac, shard.pixel_begin, shard.pixel_count)
            ev1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            d_p1 = torch.zeros((2, shard.pixel_count), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()  # (the fill runs on torch's stream; `stream` does not wait for it)
            with torch.cuda.stream(stream):
                for k in range(10):
                    eng1.process_device(d_full[k % B].data_ptr(), 1, d_p1[1].data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ev1[0].record(stream)
                for k in range(n1):
                    eng1.process_device(d_full[k % B].data_ptr(), 1, d_p1[1].data_ptr(), stream.cuda_stream)
                ev1[1].record(stream)
                torch.cuda.synchronize()
                wall1 = time.perf_counter() - t1
                eng1.process_device(d_full[0].data_ptr(), 1, d_p1[0].data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
            ms1 = ev1[0].elapsed_time(ev1[1]) / n1
            parity1 = full_grid_parity(d_p1[0].cpu().numpy(), host_first[0], off, frac, mode)
            rec = {
                "math": mode, "value": n1 / wall1, "unit": "frames/s", "calls": n1, "ms_per_frame_device": ms1,
                "kernel": pkg.binding.KERNEL_NAMES[eng1.stats().kernel_variant],
                "valu_frac": int(st.alg_flops_frame) / (ms1 * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                "hbm_frac": int(st.alg_bytes_frame) / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "parity_max_rel_err": parity1["max_rel_unfloored"], "parity": parity1,
            }
            eng1.close()
            return rec

        out["single_frame"] = single_frame_leg(args.math)
        out["single_frame"]["note"] = "one frame per call (asynchronous device-pointer entry, back to back): the regime of the reference's live path"
        out["single_frame"]["other_mode"] = single_frame_leg(other)

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
        del host_batch

        # bf16 accumulator on the same frames: its error against the fp32 sweep just run
        nf = min(B, 16)
        eng16 = job.make_engine(pkg.MATH_BF16_ACC, nf, off, frac, shard.pixel_begin, shard.pixel_count)
        d_p16 = torch.zeros((nf, shard.pixel_count), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            eng16.process_device(d_full.data_ptr(), nf, d_p16.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        p32 = d_power[:nf].cpu().numpy().astype(np.float64)
        p16 = d_p16.cpu().numpy().astype(np.float64)
        floor = 1e-4 * p32.max(axis=1, keepdims=True)
        out["bf16"] = {
            "max_rel_err": float((np.abs(p16 - p32) / np.maximum(p32, floor)).max()),
            "max_rel_err_unfloored": float((np.abs(p16 - p32) / p32).max()),
            "max_err_over_frame_peak": float((np.abs(p16 - p32) / p32.max(axis=1, keepdims=True)).max()),
            "frames": nf,
            "note": "AWPU_MATH_BF16_ACC (BASELINE configs[4], 'bf16 vs fp32 accumulator'): running sums kept in bf16 (round to "
                    "nearest even after every mic), everything else fp32.  Only its ERROR against the fp32 sweep of the same "
                    "frames is a like-for-like figure and only that is reported: the mode runs in the untuned verification "
                    "kernel's structure, so its rate says nothing about bf16 (gfx950 has no packed bf16 add; a bf16 running "
                    "sum is an fp32 add plus a convert, strictly more VALU work than the fp32 accumulator).  Percent-level "
                    "error against a 1e-5 budget: rejected (docs/HISTORY.md 8)",
        }
        eng16.close()

        out["parity_dc"] = parity_dc(pkg)
        # (advisor, round 4) the run's own mode against the reference-built goldens at every offset, beside the zero-mean parity above
        out["parity"]["dc_ok"] = all(c[f"{args.math}_ok_1e5"] for c in out["parity_dc"]["cases"])
        out["parity"]["math"] = args.math
        out["reference_default"] = reference_default(pkg, torch, args, dev, local_rank, args.math)
        out["reference_default"]["other_mode"] = reference_default(pkg, torch, args, dev, local_rank, other, cpu=False)
        # the mid-size live case: four arrays (one AWPU) on a 64 x 64 grid, one frame per call, in the run's mode
        out["single_frame_c2"] = reference_default(pkg, torch, args, dev, local_rank, args.math, cpu=False, workload="c2")
        out["workloads"] = other_workloads(pkg, sharding, torch, dist, args, dev, local_rank)
        out["projected_scaling"] = projected_scaling(pkg, sharding, torch, dist, args, spec, dev, local_rank, d_full, B,
                                                     fps_one_gpu=out["value"], ms_one_gpu=out["ms_per_step"])
        try:
            out["projected_scaling_strong"] = projected_scaling_strong(pkg, sharding, torch, dist, args, dev, local_rank)
        except Exception as exc:  # noqa: BLE001 -- an optional leg
            out["projected_scaling_strong"] = {"error": str(exc)}

    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(spec, off, frac, host_first[0], args.cpu_seconds, args.interp)
        out["cpu_baseline"]["pixels"] = int(shard.pixel_count)
        out["speedup_vs_cpu_1t"] = out["value"] / out["cpu_baseline"]["value"]  # (the run's math mode: the default unless --math says otherwise)
        if out.get("workloads") and out["workloads"][0]["workload"].startswith(spec.name):
            out["speedup_vs_cpu_1t_other_mode"] = {"math": out["workloads"][0]["math"], "value": out["workloads"][0]["value"] / out["cpu_baseline"]["value"]}
    if rank == 0:
        print(json.dumps(out), flush=True)

    # ---- N > 1, opt-in: the assembled heatmap of frame 0 must equal what the shards computed (a collective
    # after the result line, so that a rank that fails here cannot cost the run its line)
    if run_guard:  # the measured part is over: from here on nothing may turn a finished run into exit status 4
        run_guard.__exit__(None, None, None)
        run_guard = None
    if world > 1 and os.environ.get("BENCH_GATHER_CHECK", "1") == "1":  # (on by default: the RCCL schedules have only ever run over gloo)
        # (a check AFTER the result line: a collective that hangs here ends the process with status 0 and a line on stderr)
        check_guard = Watchdog(120, "gather check: timed out -- the result line above stands; the post-run heatmap gather", status=0)
        check_guard.__enter__()
        try:
            full = sharding.gather_power(d_power[:1].contiguous(), job.shards, dst=0)
            if rank == 0:
                cols = spec.res
                mine = torch.cat([full[0, r * cols:(r + 1) * cols] for r in shard.rows()]) if not c5 else full[0, : shard.pixel_count]
                ok = full.shape == (1, grid_pixels) and torch.equal(mine, d_power[0])
                msg = "ok" if ok else "MISMATCH: the assembled heatmap differs from rank 0's tile"
                if ok and not c5:  # the whole assembled heatmap against the oracle, every pixel, unfloored
                    off_all, frac_all = S.delay_table(spec, xyz, 0, spec.res)
                    rep = full_grid_parity(full[0].cpu().numpy(), host_first[0], off_all, frac_all, args.math, args.interp)
                    msg = f"{'ok' if rep['ok'] else 'PARITY FAILED'}: assembled {world}-rank heatmap vs oracle on {rep['pixels']} pixels: {rep}"
                print("gather check:", msg, file=sys.stderr)
        except Exception as exc:  # noqa: BLE001
            print(f"gather check: raised on rank {rank}: {exc}", file=sys.stderr)
        check_guard.__exit__(None, None, None)

    job.close()
    if world > 1:
        dist.destroy_process_group()


def parity_dc(pkg):
    """DC-biased input (an ADC bias: src/fpga/pipeline.cpp:290 removes none): both math modes on the committed goldens
    tests/golden/sweep_{c1,headline}_dc.npz -- hash frames + {1e-4, 1e-3, 1e-2, 0.25} through the reference's compiled
    delay() (oracle/_ref, tests/golden/make_golden.py) -- max unfloored per-pixel error against those powers, per offset.
    AWPU_MATH_F32_EXACT keeps the reference's order and stays within 1e-5 at every offset; the re-ordered
    AWPU_MATH_F32_FAST (stencil on the samples, docs/HISTORY.md 4.2g) leaves the REFERENCE's own cancellation noise behind and
    is therefore further than 1e-5 from it once the bias dwarfs the signal."""
    import util

    out = {"what": "max unfloored per-pixel error vs the powers the reference's compiled delay() produced (tests/golden/*_dc.npz), "
                   "hash frames (amplitude 2^-6) + offset; exact = AWPU_MATH_F32_EXACT (the default), fast = AWPU_MATH_F32_FAST (opt-in)",
           "cases": []}
    for name in ("sweep_c1_dc", "sweep_headline_dc"):
        g = np.load(REPO / "tests" / "golden" / f"{name}.npz")
        ax, ay = (int(v) for v in g["arrays"])
        X0 = util.hash_frames(64 * ax * ay, int(g["hist"]), seed=int(g["seed"]))[0]
        frames = np.stack([(X0 + np.float32(dc)).astype(np.float32) for dc in g["offsets"]])
        rec = {"golden": name, "mics": 64 * ax * ay, "pixels": int(g["off"].shape[0]), "offsets": [float(v) for v in g["offsets"]]}
        for label, math_id in (("exact", pkg.MATH_F32_EXACT), ("fast", pkg.MATH_F32_FAST)):
            with pkg.Engine(n_pixels=g["off"].shape[0], n_streams=frames.shape[1], lut_stride=g["off"].shape[1], hist=frames.shape[2],
                            math=math_id, max_batch=frames.shape[0]) as eng:
                eng.set_delay_table(g["off"], g["frac"])
                eng.set_active_mics(g["index"])
                power = eng.process(frames)
            rec[label] = [util.power_rel_err_unfloored(power[k], g["power"][k]) for k in range(frames.shape[0])]
        rec["exact_ok_1e5"] = bool(max(rec["exact"]) <= 1e-5)
        rec["fast_ok_1e5"] = bool(max(rec["fast"]) <= 1e-5)  # (False by design once the bias dwarfs the signal: INTEGRATION.md "Which math mode")
        out["cases"].append(rec)
    return out


def reference_default(pkg, torch, args, dev, local_rank, mode="exact", cpu=True, workload="ref_default"):
    """The configuration the reference ships and runs live: ONE 8x8 array, a 100x100 grid (src/main.cpp:38-41: --mimo-res
    100), one frame per call (MIMOWorker::update, once per 256-sample block = every 5.24 ms at 48 828 Hz).  The device time
    per frame, the reference's own delay() on one host thread beside it, and what each makes of the real-time budget."""
    S = pkg.synthetic
    spec = S.WORKLOADS[workload]
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz)
    frames = S.make_frames(xyz, 4, seed=args.seed)
    d_frames = torch.from_numpy(frames).to(dev)
    d_p = torch.zeros((2, spec.n_pixels), dtype=torch.float32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    n = 400
    math_id = pkg.MATH_F32_EXACT if mode == "exact" else pkg.MATH_F32_FAST
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=1, device=local_rank, grid_columns=spec.res, math=math_id) as eng:
        eng.set_delay_table(off, frac)
        eng.set_active_mics(None)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for k in range(20):
                eng.process_device(d_frames[k % 4].data_ptr(), 1, d_p[1].data_ptr(), stream.cuda_stream)
            ev[0].record(stream)
            for k in range(n):
                eng.process_device(d_frames[k % 4].data_ptr(), 1, d_p[1].data_ptr(), stream.cuda_stream)
            ev[1].record(stream)
            eng.process_device(d_frames[0].data_ptr(), 1, d_p[0].data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
        # one synchronous call on a host buffer: what MIMOWorker::update would wait for (upload, sweep, read-back)
        eng.process(frames[:1])
        t0 = time.perf_counter()
        for _ in range(20):
            eng.process(frames[:1])
        host_call_ms = (time.perf_counter() - t0) / 20 * 1e3
        kernel_name = pkg.binding.KERNEL_NAMES[eng.stats().kernel_variant]
    ms = ev[0].elapsed_time(ev[1]) / n
    par = full_grid_parity(d_p[0].cpu().numpy(), frames[0], off, frac, mode)
    block_ms = 256 / 48828.0 * 1e3
    out = {
        "workload": spec.name + (", one frame per call (src/main.cpp:38-41,53-56; aw_processing_unit.cpp:74)" if workload == "ref_default" else
                                 ", one frame per call (one AWPU per port, four arrays on the wire: aw_control_unit.cpp:206-213)"), "math": mode,
        "ms_per_frame_device": ms, "value": 1e3 / ms, "unit": "frames/s", "kernel": kernel_name,
        "ms_per_host_call": host_call_ms,
        "realtime_block_ms": block_ms, "fraction_of_realtime_budget": host_call_ms / block_ms,
        "parity_max_rel_unfloored": par["max_rel_unfloored"], "parity_ok": par["ok"], "pixels_checked": par["pixels"],
        "note": "device time per frame from back-to-back device-pointer calls; ms_per_host_call = awpu_hip_process on a pageable "
                "host frame (upload + sweep + read-back) through the Python binding, the call MIMOWorker::update would make every 5.24 ms "
                "(the binding adds ~5 us: the same call from C, examples/live_call_rate.c, is in profiles/r05_live_call_rate_c.txt)",
    }
    if cpu and args.cpu_seconds > 0:
        cpu = cpu_baseline(spec, off, frac, frames[0], min(3.0, args.cpu_seconds))
        out["cpu_reference_1t"] = {"value": cpu["value"], "unit": "frames/s", "ms_per_frame": 1e3 / cpu["value"],
                                   "fraction_of_realtime_budget": 1e3 / cpu["value"] / block_ms, "kind": cpu["kind"]}
    return out


def other_workloads(pkg, sharding, torch, dist, args, dev, local_rank):
    """The other BASELINE shapes in the same driver-timed run, compactly: a few timed steps each, the sweep
    launch timed by events on its stream, parity of frame 0 on every pixel."""
    S = pkg.synthetic
    out = []
    # (steps, warm-up): the short launches get more of both -- three c2 steps after one warm-up step ran 17 % below the
    # rate of a 20-step run of the same workload (profiles/r03_bench_c2.json: the clocks and the caches had not settled)
    other = "fast" if args.math == "exact" else "exact"
    # the headline in the OTHER fp32 mode first; c2 / c3 / c5 in the run's mode; FIR8 in the fast mode (its tuned kernel: the
    # reference's shipped build never compiles that variant of delay())
    for name, mode, batch, K, W in (("headline", other, 128, 4, 2), ("c2", args.math, 128, 12, 6), ("c3", args.math, 128, 4, 2),
                                    ("c3 fir8", "fast", 128, 3, 1), ("c5", args.math, 1024, 3, 1)):
        c5 = name == "c5"
        fir = name.endswith("fir8")  # BASELINE configs[2] read as the 8-tap fractional-delay variant of delay() (delay.cpp:31-40)
        exact = mode == "exact"  # AWPU_MATH_F32_EXACT: delay.cpp:19-25's operation order, mimo.cpp:124-130's mic order
        spec = S.WORKLOADS["c4" if c5 else name.split()[0]]
        t0 = time.perf_counter()
        sub = argparse.Namespace(**vars(args))
        sub.interp = "fir8" if fir else "lerp"
        sub.math = mode
        job = RankJob(pkg, sharding, torch, dist, sub, spec, 1, 0, dev, local_rank, batch, c5=c5)
        elapsed, kernel_ms, _ = job.timed(K, W)
        st = job.eng.stats()
        par = full_grid_parity(job.d_power[0].cpu().numpy(), job.host_first[0], job.off, job.frac, math=sub.math, interp=sub.interp)
        flops = int(st.alg_flops_frame) * batch
        if fir:  # 16 flop per (pixel, mic, sample) instead of 4
            flops = (16 * job.shard.pixel_count * st.usable * 256 + 6 * job.shard.pixel_count * 254) * batch
        out.append({
            "workload": (f"c5: one rank's slab of 8 of 512 mics x 256x256 ({job.shard.pixel_count} pixels), {batch} frames in flight"
                         if c5 else spec.name + (", 8-tap FIR variant of delay() (--interp fir8)" if fir else "") +
                         (", AWPU_MATH_F32_EXACT: the reference's operation and mic order (the default)" if exact else
                          ", AWPU_MATH_F32_FAST: the re-ordered fp32 sweep (--math fast, opt-in)")),
            "math": mode, "frames_per_step": batch, "steps": K, "warmup": W, "value": batch * K / elapsed, "unit": "frames/s",
            "kernel": pkg.binding.KERNEL_NAMES[st.kernel_variant],
            "kernel_ms": kernel_ms, "valu_frac": flops / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
            "hbm_frac": int(st.alg_bytes_frame) * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "parity_max_rel_unfloored": par["max_rel_unfloored"], "parity_bound": par["bound"], "parity_ok": par["ok"],
            "pixels_checked": par["pixels"], "setup_and_run_s": None,
        })
        job.close()
        del job
        torch.cuda.empty_cache()
        out[-1]["setup_and_run_s"] = round(time.perf_counter() - t0, 2)
    return out


def projected_scaling(pkg, sharding, torch, dist, args, spec, dev, local_rank, d_full, B1, fps_one_gpu, ms_one_gpu, ranks=8):
    """What ONE GPU can retire of the 8-GPU unknowns: rank 0's slab of a `ranks`-rank run (1/8 of the rows) through the
    exact N > 1 step loop -- the root's pack pass (or window cut) on the side stream, double buffer, event waits, the C-ABI
    call on its own stream --
    with the collective replaced by a local copy of the same bytes into the receive buffer (sharding.LocalCopyExchange).
    Everything a rank does per step is in the clock except the wire: xGMI and RCCL's own CU use are NOT projected."""
    B = default_batch(ranks, False) if not args.batch else args.batch
    reps = (B + d_full.shape[0] - 1) // d_full.shape[0]
    frames = d_full if reps == 1 else d_full.repeat(reps, 1, 1)[:B].contiguous()  # (the same frames again; every copy is swept)
    sub = argparse.Namespace(**vars(args))
    job = RankJob(pkg, sharding, torch, dist, sub, spec, ranks, 0, dev, local_rank, B, stub=True, frames_src=frames)
    K = 8
    elapsed, kernel_ms, t_enqueued = job.timed(K, 2)
    step_ms = elapsed / K * 1e3
    # the same slab without the exchange machinery: resident window, back-to-back launches
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    with torch.cuda.stream(job.stream):
        job.sweep(job.bufs[0])
        ev[0].record(job.stream)
        for _ in range(4):
            job.sweep(job.bufs[0])
        ev[1].record(job.stream)
        torch.cuda.synchronize()
    bare_ms = ev[0].elapsed_time(ev[1]) / 4
    # what the seven ranks that are not the ingest rank do per step: receive and sweep, no pack
    job.root_work = False
    elapsed_peer, kernel_peer_ms, _ = job.timed(K, 1)
    job.root_work = True
    exchange_mb = job.exchange_mb
    out = {
        "ranks": ranks, "rank": 0, "slab_rows": job.shard.row_count, "slab_pixels": job.shard.pixel_count,
        "rows": "row groups of four dealt round-robin" if job.shard.ranges else "contiguous slab", "exchange": job.exchange,
        "frames_per_step": B, "steps": K,
        "slab_kernel_ms": kernel_ms,            # the sweep launch inside the loop, by events on its stream
        "slab_kernel_ms_bare": bare_ms,         # the same launch back to back, nothing beside it
        "step_wall_ms": step_ms,                # wall per step of the whole loop (the root's pack or cut + stand-in copy + sweep)
        "peer_step_wall_ms": elapsed_peer / K * 1e3,  # the same for a rank other than the ingest rank (no pack / cut)
        "peer_slab_kernel_ms": kernel_peer_ms,
        "host_enqueue_ms_per_step": t_enqueued / K * 1e3,  # Python + ctypes + event records: hidden while < step_wall_ms
        "host_overhead_ms": max(0.0, step_ms - kernel_ms),  # what the loop adds to the kernel on the device's timeline
        "exchange_mb_per_step": exchange_mb,
        "exchange_gbs_needed": exchange_mb / 1e3 / (step_ms * 1e-3),  # what RCCL must deliver to every rank to stay hidden
        "projected_value": B / (step_ms * 1e-3), "unit": "frames/s",
        "ceiling_x": (B / (step_ms * 1e-3)) / fps_one_gpu,
        "one_gpu_ms_per_step": ms_one_gpu, "one_gpu_frames_per_step": B1,
        "note": f"rank 0's slab of {ranks} on ONE GPU, collective replaced by a local device copy of the same {exchange_mb:.0f} MB: an upper "
                f"bound on the {ranks}-GPU value from everything but the wire (xGMI bandwidth, RCCL's CU use and cross-rank skew "
                f"are not in it).  NOT a scaling measurement",
    }
    job.close()
    del job
    # the same share under the raw_scatter schedule (sharding.RawScatterExchange): no pack pass of the whole batch on the
    # ingest rank -- every rank receives its eighth of the raw snapshots and packs that; ingest rank and peers then do the same
    try:
        job2 = RankJob(pkg, sharding, torch, dist, sub, spec, ranks, 0, dev, local_rank, B, stub="raw_scatter", frames_src=frames)
        if job2.raw_ok:
            e2, k2, _ = job2.timed(K, 2)
            out["raw_scatter"] = {
                "step_wall_ms": e2 / K * 1e3, "slab_kernel_ms": k2, "ceiling_x": (B / (e2 / K)) / fps_one_gpu,
                "raw_mb_per_rank_per_step": (B // ranks) * spec.n_mics * pkg.binding.HIST * 4 / 1e6,
                "note": "every rank (the ingest rank too) under the raw_scatter schedule: its eighth of the raw snapshots arrives by a "
                        "local copy, it packs them into its slot, the other slots arrive packed; bench.py --gpus N times this schedule "
                        "against the other two on the node and keeps the fastest",
            }
        job2.close()
    except Exception as exc:  # noqa: BLE001 -- an optional leg
        out["raw_scatter"] = {"error": str(exc)}
    return out


def projected_scaling_strong(pkg, sharding, torch, dist, args, dev, local_rank, ranks=8, batch=128):
    """BASELINE configs[3] -- 512 mics, ONE 256x256 frame stream sharded over 8 GPUs -- is the STRONG-scaling case: the batch a
    display wants per step is fixed (`batch` frames) and every GPU sweeps an eighth of the grid for all of them.  What one
    GPU can say about it: the whole c4 grid at that batch on this GPU (the N = 1 step), then rank 0's share of an 8-rank run
    through the exact N > 1 loop with the collective replaced by a local copy of the same bytes (as projected_scaling).
    ceiling_x = one-GPU step / rank-0 step: everything but the wire.  NOT a scaling measurement."""
    S = pkg.synthetic
    spec = S.WORKLOADS["c4"]
    sub = argparse.Namespace(**vars(args))
    sub.batch = batch  # (a fixed batch: "scaling": "strong")
    one = RankJob(pkg, sharding, torch, dist, sub, spec, 1, 0, dev, local_rank, batch)
    e1, k1, _ = one.timed(3, 1)
    frames = one.d_full
    one_ms = e1 / 3 * 1e3
    one.eng.close()
    job = RankJob(pkg, sharding, torch, dist, sub, spec, ranks, 0, dev, local_rank, batch, stub=True, frames_src=frames)
    K = 8
    e8, k8, _ = job.timed(K, 2)
    step_ms = e8 / K * 1e3
    job.root_work = False
    ep, kp, _ = job.timed(K, 1)
    out = {
        "workload": f"configs[3]: {spec.name}, {batch} frames per step whatever the number of GPUs (strong scaling)",
        "ranks": ranks, "frames_per_step": batch, "slab_rows": job.shard.row_count, "exchange": job.exchange,
        "one_gpu_ms_per_step": one_ms, "one_gpu_kernel_ms": k1, "one_gpu_value": batch / (one_ms * 1e-3),
        "step_wall_ms": step_ms, "slab_kernel_ms": k8, "peer_step_wall_ms": ep / K * 1e3,
        "exchange_mb_per_step": job.exchange_mb, "exchange_gbs_needed": job.exchange_mb / 1e3 / (step_ms * 1e-3),
        "projected_value": batch / (step_ms * 1e-3), "unit": "frames/s", "ceiling_x": one_ms / step_ms,
        "note": f"rank 0's eighth of the 256x256 grid on ONE GPU, the frame exchange replaced by a local device copy of the same "
                f"{job.exchange_mb:.0f} MB: an upper bound on the {ranks}-GPU value from everything but the wire.  NOT a scaling measurement",
    }
    job.close()
    del job, one, frames
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
