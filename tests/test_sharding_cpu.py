"""The multi-GPU decomposition on CPU: row-slab planning, the double-buffered frame broadcast and
the tile gather, with world_size 2 over gloo.  The sweep itself needs a GPU, so each rank's slab is
computed by the oracle here (tests may use it; the product path never does) -- what is under test
is that shards tile the grid, that every rank receives rank 0's frames, and that the assembled
heatmap equals the single-rank one."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_rows_tile_the_grid(pkg):
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    for res, world in [(128, 8), (100, 8), (32, 3), (7, 7), (256, 1)]:
        shards = sh.all_shards(res, res, world)
        assert shards[0].row_begin == 0 and shards[-1].row_begin + shards[-1].row_count == res
        for a, b in zip(shards, shards[1:]):
            assert a.row_begin + a.row_count == b.row_begin
        counts = [s.row_count for s in shards]
        assert max(counts) - min(counts) <= 1 and sum(s.pixel_count for s in shards) == res * res
    with pytest.raises(ValueError):
        sh.shard_rows(4, 4, 8, 0)


def _worker(rank, world, port, res, batch, steps, out_dir, mode="broadcast", p2p=None):
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(repo))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("beamforming-lk_amd")
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    from oracle import oracle_py

    xyz = oracle_py.create_antenna()
    off, frac = oracle_py.compute_delay_lut(xyz, res, res)
    shard = sh.shard_rows(res, res, world, rank)
    sl = slice(shard.pixel_begin, shard.pixel_begin + shard.pixel_count)
    rng = np.random.default_rng(100)
    bufs = tuple(torch.zeros((batch, 64, 1024), dtype=torch.float32) for _ in range(2))
    bc = sh.FrameBroadcaster(bufs, src=0, mode=mode, point_to_point=p2p)
    assert bc.active and bc.mode == mode
    results = []
    all_frames = [rng.uniform(-0.01, 0.01, size=(batch, 64, 1024)).astype(np.float32) for _ in range(steps)]
    if rank == 0:
        bufs[0].copy_(torch.from_numpy(all_frames[0]))
    bc.post(0)
    for k in range(steps):
        frames = bc.wait(k)
        if k + 1 < steps:
            if rank == 0:
                bufs[(k + 1) % 2].copy_(torch.from_numpy(all_frames[k + 1]))
            bc.post(k + 1)
        got = frames.numpy()
        assert np.array_equal(got, all_frames[k]), f"rank {rank} step {k}: frames differ from rank 0's"
        local = np.stack([oracle_py.das_f32(got[b], off[sl], frac[sl]) for b in range(batch)])
        full = sh.gather_power(torch.from_numpy(local), sh.all_shards(res, res, world), dst=0)
        if rank == 0:
            results.append(full.numpy())
        else:
            assert full is None
        # display step across tiles: all-reduce(MAX) of the tile maxima, then every rank scales
        # its tile alike; the assembled image must be the single-process populateHeatmap of the full grid
        peak = sh.global_peak(torch.from_numpy(local.max(axis=1).copy()))
        whole = np.stack([oracle_py.das_f32(got[b], off, frac) for b in range(batch)])
        assert np.array_equal(peak.numpy(), whole.max(axis=1))
        for b in range(batch):  # the host heatmap takes its own maximum: hand it the global one as an extra pixel
            tile = pkg.heatmap_u8(np.concatenate([local[b], peak.numpy()[b:b + 1]]))[:-1]
            assert np.array_equal(tile, oracle_py.heatmap_u8(whole[b])[sl])
    if rank == 0:
        want = np.stack([np.stack([oracle_py.das_f32(all_frames[k][b], off, frac) for b in range(batch)]) for k in range(steps)])
        np.save(os.path.join(out_dir, "ok.npy"), np.array([np.array_equal(np.stack(results), want)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,res,mode", [(2, 10, "broadcast"), (2, 7, "broadcast"), (2, 6, "scatter_allgather")])
def test_broadcast_and_gather_world2(tmp_path, world, res, mode):
    port = free_port()
    mp.spawn(_worker, args=(world, port, res, 2, 3, str(tmp_path), mode), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0]


@pytest.mark.parametrize("world,res,batch", [(4, 8, 4), (3, 9, 6)])
def test_rccl_exchange_schedule_over_gloo(tmp_path, world, res, batch):
    """The schedule the N > 1 bench runs over RCCL -- the root's slices sent point to point straight into every peer's
    batch buffer, then the in-place all-gather, ordered after the receives -- on more than two ranks, over gloo:
    every rank ends up with rank 0's frames, three steps through the double buffer."""
    port = free_port()
    mp.spawn(_worker, args=(world, port, res, batch, 3, str(tmp_path), "scatter_allgather", True), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0]


def _raw_scatter_worker(rank, world, port, steps, out_dir):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    batch, mics, hist, wp = 4 * world, 3, 16, 5
    per = batch // world
    rng = np.random.default_rng(5)
    all_frames = [rng.standard_normal((batch, mics, hist)).astype(np.float32) for _ in range(steps)]

    def pack(raw, slot):  # a stand-in for awpu_hip_pack_frames: two frames interleaved, a window of wp samples from sample 2
        slot.copy_(torch.stack([raw[0::2, :, 2:2 + wp], raw[1::2, :, 2:2 + wp]], dim=-1).reshape(slot.shape))

    packed = tuple(torch.zeros((batch // 2, mics * wp * 2)) for _ in range(2))
    raw = tuple(torch.zeros((per, mics, hist)) for _ in range(2)) if rank != 0 else None
    full = [torch.from_numpy(f) for f in all_frames]
    ex = sh.RawScatterExchange(packed, raw, (lambda k: full[k]) if rank == 0 else None, pack, src=0)
    ok = True
    ex.post(0)
    for k in range(steps):
        got = ex.wait(k)
        if k + 1 < steps:
            ex.post(k + 1)
        want = np.stack([all_frames[k][0::2, :, 2:2 + wp], all_frames[k][1::2, :, 2:2 + wp]], axis=-1).reshape(batch // 2, -1)
        ok &= np.array_equal(got.numpy(), want)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.save(os.path.join(out_dir, "ok.npy"), flag.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_raw_scatter_exchange(tmp_path, world):
    """RawScatterExchange: raw slices to every rank, every rank packs its own, in-place all-gather of the packed batch --
    every rank ends up with the whole packed batch, three steps through the double buffer."""
    port = free_port()
    mp.spawn(_raw_scatter_worker, args=(world, port, 3, str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0]


def _scatter_worker(rank, world, port, steps, out_dir):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    batch, per = 6, 6 // world
    first, count = sh.shard_frames(batch, world, rank)
    assert count == per
    rng = np.random.default_rng(7)
    all_frames = [rng.standard_normal((batch, 4, 32)).astype(np.float32) for _ in range(steps)]
    local = tuple(torch.zeros((per, 4, 32)) for _ in range(2))
    full = tuple(torch.zeros((batch, 4, 32)) for _ in range(2)) if rank == 0 else None
    sc = sh.FrameScatterer(local, full, src=0)
    if rank == 0:
        full[0].copy_(torch.from_numpy(all_frames[0]))
    sc.post(0)
    ok = True
    for k in range(steps):
        mine = sc.wait(k)
        if k + 1 < steps:
            if rank == 0:
                full[(k + 1) % 2].copy_(torch.from_numpy(all_frames[k + 1]))
            sc.post(k + 1)
        ok &= np.array_equal(mine.numpy(), all_frames[k][first:first + count])
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.save(os.path.join(out_dir, "ok.npy"), flag.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_frame_scatter_world2(tmp_path):
    """The frame-sharded decomposition: every rank receives exactly its frames of rank 0's batches."""
    mp.spawn(_scatter_worker, args=(2, free_port(), 4, str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "ok.npy")[0] == 1


def test_shard_frames_is_a_balanced_partition(pkg):
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    for batch, world in [(128, 8), (10, 4), (3, 8), (64, 1)]:
        parts = [sh.shard_frames(batch, world, r) for r in range(world)]
        assert parts[0][0] == 0 and sum(n for _, n in parts) == batch
        for (a, n), (b, _) in zip(parts, parts[1:]):
            assert a + n == b
        assert max(n for _, n in parts) - min(n for _, n in parts) <= 1


def test_broadcaster_is_a_noop_without_a_group(pkg):
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    bufs = (torch.ones(3), torch.zeros(3))
    bc = sh.FrameBroadcaster(bufs)
    assert not bc.active
    bc.post(0)
    assert bc.wait(0) is bufs[0] and bc.wait(1) is bufs[1]
    t = torch.arange(6.0).reshape(2, 3)
    assert sh.gather_power(t, sh.all_shards(3, 1, 1)) is t
    assert sh.global_peak(t) is t


def test_interleaved_row_groups_tile_the_grid(pkg):
    """shard_rows_interleaved: row groups of four dealt round-robin -- every row owned once, quads intact (every range
    is a whole group of four adjacent rows), equal shares, and with a cost that grows linearly from the centre row
    outwards (what the sweep shows) every rank's cost is the same; assemble_tiles puts the tiles back in grid order."""
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    for res, world in [(128, 8), (128, 4), (128, 2), (256, 8), (64, 8), (100, 8), (30, 4), (128, 1)]:
        shards = sh.all_shards(res, res, world, interleaved=True)
        rows = sorted(r for s in shards for r in s.rows())
        assert rows == list(range(res)), (res, world)
        for s in shards:
            assert s.row_count == len(s.rows()) and s.pixel_count == s.row_count * res
            for b, n in s.row_ranges[:-1]:
                assert world == 1 or (b % 4 == 0 and n == 4)
        if res % (4 * world) == 0 and world > 1:
            cost = [sum(abs(r - (res - 1) / 2) for r in s.rows()) for s in shards]
            assert max(cost) - min(cost) < 1e-9, (res, world, cost)
            contiguous = [sum(abs(r - (res - 1) / 2) for r in s.rows()) for s in sh.all_shards(res, res, world)]
            assert world == 2 or max(contiguous) > 1.5 * min(contiguous)
    assert sh.shard_rows_interleaved(8, 8, 4, 1).row_ranges == ((2, 2),)  # fewer groups than ranks: contiguous slabs
    shards = sh.all_shards(16, 3, 2, interleaved=True)
    image = torch.arange(2 * 16 * 3, dtype=torch.float32).reshape(2, 48)
    tiles = [torch.cat([image[:, r * 3:(r + 1) * 3] for r in s.rows()], dim=1) for s in shards]
    assert torch.equal(sh.assemble_tiles(tiles, shards), image)


def test_local_copy_exchange_follows_the_broadcaster_protocol(pkg):
    """bench.py's projected_scaling stand-in for the collective: post(k) delivers `arrival` into buffer k % 2, wait(k)
    hands that buffer out -- the FrameBroadcaster protocol, so the N > 1 step loop runs unchanged on one device."""
    sh = importlib.import_module("beamforming-lk_amd.sharding")
    bufs = tuple(torch.zeros((3, 4, 8)) for _ in range(2))
    arrival = torch.arange(3 * 4 * 8, dtype=torch.float32).reshape(3, 4, 8)
    ex = sh.LocalCopyExchange(bufs, arrival)
    assert ex.mode == "local_copy"
    ex.post(0)
    for k in range(5):
        got = ex.wait(k)
        assert got is bufs[k % 2] and torch.equal(got, arrival)
        got.zero_()  # (the sweep may do what it likes with a buffer it was handed)
        if k + 1 < 5:
            arrival += 1.0
            ex.post(k + 1)
    with pytest.raises(ValueError):
        sh.LocalCopyExchange(bufs, torch.zeros((2, 4, 8)))
