"""Multi-GPU decomposition of the heatmap sweep: one process per GPU, each owning a slab of
grid rows; the frame batch is broadcast from the ingest rank once per step (RCCL over xGMI when
the backend is "nccl", gloo on CPU for the tests); no other collective is on the data path.

The reference is single-process (one MIMOWorker thread, src/dsp/mimo.cpp:12) -- this module has
no counterpart there.  Pixels are independent until display-time normalisation
(src/dsp/mimo.cpp:61-95), so the grid shards with no halo.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class RowShard:
    rank: int
    world: int
    res_rows: int
    res_cols: int
    row_begin: int   # first grid row owned (of the first range when the shard is interleaved)
    row_count: int   # rows owned in total
    ranges: Tuple[Tuple[int, int], ...] = ()  # interleaved shards: (first row, rows) of every range, ascending

    @property
    def pixel_begin(self) -> int:
        return self.row_begin * self.res_cols

    @property
    def pixel_count(self) -> int:
        return self.row_count * self.res_cols

    @property
    def row_ranges(self) -> Tuple[Tuple[int, int], ...]:
        """(first grid row, rows) of every contiguous run of rows this shard owns, in the order its tile holds them."""
        return self.ranges if self.ranges else ((self.row_begin, self.row_count),)

    def rows(self) -> List[int]:
        return [r for b, n in self.row_ranges for r in range(b, b + n)]


def shard_rows(res_rows: int, res_cols: int, world: int, rank: int) -> RowShard:
    """Contiguous, balanced row slabs: the first (res_rows % world) ranks get one extra row."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    if world > res_rows:
        raise ValueError("more ranks than grid rows")
    base, extra = divmod(res_rows, world)
    begin = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return RowShard(rank, world, res_rows, res_cols, begin, count)


def all_shards(res_rows: int, res_cols: int, world: int, interleaved: bool = False) -> List[RowShard]:
    f = shard_rows_interleaved if interleaved else shard_rows
    return [f(res_rows, res_cols, world, r) for r in range(world)]


def shard_rows_interleaved(res_rows: int, res_cols: int, world: int, rank: int, group: int = 4) -> RowShard:
    """Row slabs dealt round-robin in groups of `group` rows: rank r owns row groups r, r + world, r + 2 world, ...

    Why: the sweep's cost per row is not flat over the grid.  The quad shape shares work between four vertically
    adjacent pixels wherever their integer delays coincide, and they coincide less towards the edge of the sine-space
    grid (headline shape, 8 contiguous slabs of 16 rows, same GPU: 5.45 ms for an edge slab against 4.93 ms for a
    middle one; an N-GPU step takes as long as its slowest rank).  Dealt round-robin every rank gets edge and centre
    groups alike -- with a cost that grows linearly from the centre row outwards the shares are exactly equal -- and
    a group of four rows is the quad shape's unit, so every quad still is four truly adjacent grid rows.  The rank's
    table and power tile hold its groups in ascending order; `row_ranges` says which grid rows they are.
    Falls back to contiguous slabs when there are fewer row groups than ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    n_groups = (res_rows + group - 1) // group
    if world == 1 or n_groups < world:
        return shard_rows(res_rows, res_cols, world, rank)
    ranges = tuple((g * group, min(group, res_rows - g * group)) for g in range(rank, n_groups, world))
    return RowShard(rank, world, res_rows, res_cols, ranges[0][0], sum(n for _, n in ranges), ranges)


class FrameBroadcaster:
    """Double-buffered broadcast of frame batches from `src` to every rank.

    post(k) starts the broadcast of batch k into buffer k % 2 (async); wait(k) returns that
    buffer once the batch has landed.  The sweep of batch k runs while batch k+1 is in flight.
    With world size 1 (or no process group) it degenerates to handing out the local buffer.

    mode "broadcast": one dist.broadcast per batch (a ring on xGMI: the root's outgoing link sets
    the rate).  mode "scatter_allgather": the root scatters 1/world of the batch to every rank over
    its point-to-point links, then an all-gather completes every rank's copy -- the bandwidth-optimal
    broadcast on a fully connected mesh, one more collective launch per batch.  The batch dimension
    must divide by the world size for it; otherwise it falls back to "broadcast".
    """

    def __init__(self, buffers: Tuple[torch.Tensor, torch.Tensor], src: int = 0,
                 group: Optional[dist.ProcessGroup] = None, mode: str = "broadcast", point_to_point: Optional[bool] = None):
        # point_to_point: the scatter step as grouped sends / receives straight into the batch buffer + an in-place
        # all-gather (what runs over RCCL).  None = by backend (gloo: scatter into a side buffer + out-of-place gather);
        # True lets the CPU tests run the RCCL schedule over gloo.
        self.point_to_point = point_to_point
        self.buffers = buffers
        self.src = src
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self._work = [None, None]
        self._mine = [None, None]
        self.mode = mode
        if self.active:
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            if mode == "scatter_allgather" and buffers[0].shape[0] % self.world != 0:
                self.mode = "broadcast"
        if mode not in ("broadcast", "scatter_allgather"):
            raise ValueError("mode must be 'broadcast' or 'scatter_allgather'")

    def post(self, k: int) -> None:
        if not self.active:
            return
        buf = self.buffers[k % 2]
        if self.mode == "broadcast":
            self._work[k % 2] = [dist.broadcast(buf, src=self.src, group=self.group, async_op=True)]
            return
        per = buf.shape[0] // self.world
        p2p = self.point_to_point if self.point_to_point is not None else dist.get_backend(self.group) != "gloo"
        if p2p:
            # NCCL/RCCL: the root sends every peer its slice straight into that peer's batch buffer (grouped point-to-point
            # sends), then the in-place all-gather (input = the output's own slice): no staging copy on either side,
            # stream-ordered one after the other
            mine = buf[self.rank * per:(self.rank + 1) * per]
            if self.rank == self.src:  # the root's own slice is in place already; one send per peer, each over its own link
                ops = [dist.P2POp(dist.isend, buf[r * per:(r + 1) * per], r, self.group) for r in range(self.world) if r != self.src]
            else:
                ops = [dist.P2POp(dist.irecv, mine, self.src, self.group)]
            works = list(dist.batch_isend_irecv(ops))
            # The all-gather must not start before this rank's slice has arrived.  Batched point-to-point ops run on the
            # collectives' stream in current PyTorch, but that is an implementation detail: wait() here only orders the
            # calling stream (the side stream) after them -- no host block -- and the all-gather below is ordered after
            # the calling stream.
            for w in works:
                w.wait()
            gather = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True)
            if dist.get_backend(self.group) == "gloo":
                # (tests: gloo runs concurrent collectives on several threads in any order, and a second wait() on one of
                # its finished send / receive works never returns: finish here, keep nothing)
                gather.wait()
                works = []
            else:
                works.append(gather)
            self._work[k % 2] = works
            return
        if self._mine[k % 2] is None:  # this rank's 1/world of the batch, outside `buf` (gloo does not gather in place)
            self._mine[k % 2] = torch.empty_like(buf[:per])
        mine = self._mine[k % 2]
        chunks = [buf[r * per:(r + 1) * per].contiguous() for r in range(self.world)] if self.rank == self.src else None
        w1 = dist.scatter(mine, chunks, src=self.src, group=self.group, async_op=True)
        w1.wait()  # gloo runs async ops on several threads, not in issue order (NCCL/RCCL is stream-ordered)
        w2 = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True)
        self._work[k % 2] = [w1, w2]

    def wait(self, k: int) -> torch.Tensor:
        works = self._work[k % 2]
        if works is not None:
            for w in works:
                w.wait()  # on CUDA this orders the current stream after the collective
            self._work[k % 2] = None
        return self.buffers[k % 2]


class RawScatterExchange:
    """The third way to get a batch to every rank: spread the ROOT's pack pass over the ranks.

    post(k): the root sends rank r the r-th slice of the RAW frames (contiguous snapshots, no window cut, no pack pass on
    the root: its share of a step is then the share of every other rank), every rank packs its slice into its slot of
    packed buffer k % 2 (`pack(raw frames [per, ...], packed slot [per / 2, ...])`: awpu_hip_pack_frames on the calling
    stream) and an in-place all-gather completes every rank's packed buffer.  wait(k) returns that buffer.
    Against FrameBroadcaster's scatter + all-gather the wire carries the raw snapshots once more (1024 / W times the
    window's bytes to each peer), which is why bench.py times the schedules against each other on the node before it
    picks one.  The batch must divide by twice the world size (whole frame pairs per rank).

    raw: this rank's two receive buffers [per, ...] (the root needs none: it packs from `full` in place);
    full: the root's whole batch of raw frames [batch, ...], or a callable k -> that tensor."""

    mode = "raw_scatter"

    def __init__(self, packed: Tuple[torch.Tensor, torch.Tensor], raw: Optional[Tuple[torch.Tensor, torch.Tensor]], full, pack,
                 src: int = 0, group: Optional[dist.ProcessGroup] = None):
        self.packed, self.raw, self.full, self.pack, self.src, self.group = packed, raw, full, pack, src, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        pairs = packed[0].shape[0]
        if pairs % self.world:
            raise ValueError("the frame pairs of a batch must divide by the world size")
        self.pairs_per = pairs // self.world
        self.per = 2 * self.pairs_per
        if self.rank != src and (raw is None or raw[0].shape[0] != self.per):
            raise ValueError("every rank but the source needs two raw buffers of batch / world frames")
        self._work = [None, None]

    def post(self, k: int) -> None:
        b = k % 2
        gloo = dist.get_backend(self.group) == "gloo"
        if self.rank == self.src:
            full = self.full(k) if callable(self.full) else self.full
            ops = [dist.P2POp(dist.isend, full[r * self.per:(r + 1) * self.per], r, self.group) for r in range(self.world) if r != self.src]
            mine_raw = full[self.src * self.per:(self.src + 1) * self.per]
        else:
            ops = [dist.P2POp(dist.irecv, self.raw[b], self.src, self.group)]
            mine_raw = self.raw[b]
        works = list(dist.batch_isend_irecv(ops)) if ops else []
        if self.rank != self.src:  # (the pack below reads what the receive delivers: order the calling stream after it)
            for w in works:
                w.wait()
            if gloo:
                works = []
        slot = self.packed[b][self.rank * self.pairs_per:(self.rank + 1) * self.pairs_per]
        self.pack(mine_raw, slot)
        gather = dist.all_gather_into_tensor(self.packed[b], slot, group=self.group, async_op=True)
        if gloo:  # (tests: see FrameBroadcaster.post)
            gather.wait()
            for w in works:
                w.wait()
            works = []
        else:
            works.append(gather)
        self._work[b] = works

    def wait(self, k: int) -> torch.Tensor:
        works = self._work[k % 2]
        if works:
            for w in works:
                w.wait()
        self._work[k % 2] = None
        return self.packed[k % 2]


class LocalCopyExchange:
    """Stand-in for FrameBroadcaster on ONE device (bench.py's projected_scaling): the same post/wait protocol and
    the same stream ordering as an asynchronous collective -- the transfer waits for what the caller's stream has
    enqueued so far, runs on a side stream, and wait() orders the caller's stream after it -- but the bytes come
    from `arrival`, a local tensor of the batch's shape, instead of over the wire.  It measures everything a rank's
    step loop costs except the wire itself."""

    mode = "local_copy"

    def __init__(self, buffers: Tuple[torch.Tensor, torch.Tensor], arrival: torch.Tensor, priority: int = 0, raw_scatter=None):
        # raw_scatter = (raw receive buffer, raw source of the same shape, pack(raw, slot), rank, world): stand in for
        # RawScatterExchange instead -- the rank's raw slice "arrives" by a local copy, the rank packs it into its slot, the
        # other ranks' slots "arrive" from `arrival`
        if arrival.shape != buffers[0].shape:
            raise ValueError("arrival must have the shape of a batch buffer")
        self.raw_scatter = raw_scatter
        if raw_scatter is not None:
            self.mode = "local_copy of raw_scatter"
        self.buffers = buffers
        self.arrival = arrival
        self.cuda = buffers[0].is_cuda
        if self.cuda:
            self.side = torch.cuda.Stream(device=buffers[0].device, priority=priority)
            self.posted = [torch.cuda.Event(), torch.cuda.Event()]
            self.landed = [torch.cuda.Event(), torch.cuda.Event()]
        self._pending = [False, False]

    def post(self, k: int) -> None:
        b = k % 2
        if self.cuda:
            self.posted[b].record(torch.cuda.current_stream(self.buffers[b].device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.posted[b])
                if self.raw_scatter is None:
                    self.buffers[b].copy_(self.arrival, non_blocking=True)
                else:
                    raw, raw_src, pack, rank, world = self.raw_scatter
                    per = self.buffers[b].shape[0] // world
                    raw.copy_(raw_src, non_blocking=True)
                    pack(raw, self.buffers[b][rank * per:(rank + 1) * per])
                    if rank > 0:
                        self.buffers[b][:rank * per].copy_(self.arrival[:rank * per], non_blocking=True)
                    if rank + 1 < world:
                        self.buffers[b][(rank + 1) * per:].copy_(self.arrival[(rank + 1) * per:], non_blocking=True)
                self.landed[b].record(self.side)
        else:
            self.buffers[b].copy_(self.arrival)
        self._pending[b] = True

    def wait(self, k: int) -> torch.Tensor:
        b = k % 2
        if self._pending[b] and self.cuda:
            torch.cuda.current_stream(self.buffers[b].device).wait_event(self.landed[b])
        self._pending[b] = False
        return self.buffers[b]


def global_peak(local_peak: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Display-time normalisation across tiles: MIMOWorker::populateHeatmap scales by the maximum over the
    WHOLE grid (src/dsp/mimo.cpp:62-73), so every rank's per-frame tile maximum [batch] is all-reduced
    (MAX, in place; one float per frame) before the tiles are scaled with peak_given (awpu_hip_heatmap_u8_device)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_peak, op=dist.ReduceOp.MAX, group=group)
    return local_peak


def shard_frames(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """The other decomposition: whole frames per rank (frame k of a batch goes to one GPU, which sweeps
    the full grid).  Throughput-only -- a frame's latency is that of one GPU -- but each frame crosses
    xGMI once instead of world-1 times.  Returns (first frame, frame count) of this rank's contiguous,
    balanced share."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, extra = divmod(batch, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


class FrameScatterer:
    """Double-buffered scatter for the frame-sharded decomposition (shard_frames): post(k) sends slice r
    of the source's full batch buffer k % 2 to rank r's local buffer k % 2 (async); wait(k) returns the
    local buffer once it has landed.  The batch must divide by the world size.  Without a process
    group the local buffers are handed out as they are."""

    def __init__(self, local: Tuple[torch.Tensor, torch.Tensor], full: Optional[Tuple[torch.Tensor, torch.Tensor]],
                 src: int = 0, group: Optional[dist.ProcessGroup] = None):
        self.local = local
        self.full = full
        self.src = src
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self._work = [None, None]
        if self.active:
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            if self.rank == src and (full is None or full[0].shape[0] != local[0].shape[0] * self.world):
                raise ValueError("the source needs full buffers of world x the local batch")

    def post(self, k: int) -> None:
        if not self.active:
            return
        chunks = None
        if self.rank == self.src:
            per = self.local[0].shape[0]
            chunks = [self.full[k % 2][r * per:(r + 1) * per] for r in range(self.world)]  # contiguous slices
        self._work[k % 2] = dist.scatter(self.local[k % 2], chunks, src=self.src, group=self.group, async_op=True)

    def wait(self, k: int) -> torch.Tensor:
        if self._work[k % 2] is not None:
            self._work[k % 2].wait()
            self._work[k % 2] = None
        return self.local[k % 2]


def gather_power(local: torch.Tensor, shards: List[RowShard], dst: int = 0,
                 group: Optional[dist.ProcessGroup] = None) -> Optional[torch.Tensor]:
    """Assemble the [batch, P] heatmap on `dst` from per-rank [batch, pixel_count] tiles (contiguous or interleaved
    shards: every tile's rows go where `row_ranges` says).  P = the pixels the shards cover, in grid order.
    Not on the timed data path (each rank can hand its tile to its own consumer)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    rank = dist.get_rank(group)
    batch = local.shape[0]
    widest = max(s.pixel_count for s in shards)
    padded = torch.zeros((batch, widest), dtype=local.dtype, device=local.device)
    padded[:, : local.shape[1]] = local
    tiles = [torch.empty_like(padded) for _ in shards] if rank == dst else None
    dist.gather(padded, tiles, dst=dst, group=group)
    if rank != dst:
        return None
    return assemble_tiles([t[:, : s.pixel_count] for t, s in zip(tiles, shards)], shards)


def assemble_tiles(tiles: List[torch.Tensor], shards: List[RowShard]) -> torch.Tensor:
    """[batch, covered pixels] in grid-row order from the shards' tiles."""
    cols = shards[0].res_cols
    owner = sorted((row, k, i) for k, s in enumerate(shards) for i, row in enumerate(s.rows()))
    batch = tiles[0].shape[0]
    out = torch.empty((batch, len(owner) * cols), dtype=tiles[0].dtype, device=tiles[0].device)
    for pos, (_, k, i) in enumerate(owner):
        out[:, pos * cols:(pos + 1) * cols] = tiles[k][:, i * cols:(i + 1) * cols]
    return out
