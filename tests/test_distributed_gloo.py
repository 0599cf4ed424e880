"""world_size-2 gloo test of the N > 1 path (CPU): stream sharding + the velocity all_gather."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import GOLDEN  # noqa: F401
from evfly_amd.distributed import gather_velocities, shard_row_counts, shard_streams


def _worker(rank, world, port, n_streams, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s0, s1 = shard_streams(n_streams, rank, world)
    # velocity rows of this rank's streams: row value encodes (stream, t) so order is checkable
    vel = torch.tensor([[s, t, s * 100 + t] for s in range(s0, s1) for t in range(T)], dtype=torch.float32).reshape(-1, 3)
    out = gather_velocities(vel, dist)
    q.put((rank, out.tolist()))          # plain lists: no shared-memory handle that dies with this process
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_streams", [8, 7])          # even and ragged shards
def test_shard_and_gather(n_streams):
    world, T = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_streams) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.tensor([[s, t, s * 100 + t] for s in range(n_streams) for t in range(T)], dtype=torch.float32)
    for r in range(world):
        assert torch.equal(torch.tensor(outs[r]), want)


def _rows(s0, s1, T):
    return torch.tensor([[s, t, s * 100 + t] for s in range(s0, s1) for t in range(T)], dtype=torch.float32).reshape(-1, 3)


def _worker_twice(rank, world, port, T, q):
    """Two gathers in ONE process group: even shards (64 streams), then ragged ones (65) in which rank 1 keeps the row
    count it had before while rank 0's grows -- the case a cache keyed on rank-local data deadlocks on; then the same
    two again with the counts supplied by the caller (the benchmark's form: one collective per call)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for n_streams, explicit in ((64, False), (65, False), (64, True), (65, True)):
        s0, s1 = shard_streams(n_streams, rank, world)
        counts = shard_row_counts(n_streams, world, T) if explicit else None
        res.append(gather_velocities(_rows(s0, s1, T), dist, counts=counts).tolist())
    try:
        gather_velocities(_rows(0, 3, T), dist, counts=[1] * world)
        res.append("no error")
    except ValueError:
        res.append("ValueError")
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_twice_even_then_ragged_same_group():
    world, T = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_twice, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for got, n_streams in zip(outs[r][:4], (64, 65, 64, 65)):
            assert torch.equal(torch.tensor(got), _rows(0, n_streams, T))
        assert outs[r][4] == "ValueError"


def _worker_c4(rank, world, port, q):
    """BASELINE config C4's shape on 4 ranks: 2048 streams x 5 windows sharded over 8 GPUs -- here ranks 0..3 of a 4-rank group
    each hold the rows two of those GPUs would (512 streams = 2560 velocity rows per rank), counts from `shard_row_counts`
    (no exchange), ONE all_gather_into_tensor per call, three calls in a row like the benchmark's steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_streams, T = 2048, 5
    counts = shard_row_counts(n_streams, world, T)
    s0, s1 = shard_streams(n_streams, rank, world)
    ok = counts == [512 * T] * world and (s0, s1) == (512 * rank, 512 * (rank + 1))
    base = torch.arange(s0 * T, s1 * T, dtype=torch.float32)[:, None]
    for step in range(3):
        vel = torch.cat([base, base * 0.5 + step, -base], 1)
        out = gather_velocities(vel, dist, counts=counts)
        want = torch.arange(n_streams * T, dtype=torch.float32)[:, None]
        ok = ok and out.shape == (n_streams * T, 3) and torch.equal(out, torch.cat([want, want * 0.5 + step, -want], 1))
    # max-over-ranks timing reduction of the benchmark (gloo has MAX)
    tm = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    every = torch.empty(world, dtype=torch.float64)
    dist.all_gather_into_tensor(every, torch.tensor([float(rank + 1)], dtype=torch.float64))
    ok = ok and tm.item() == world and every.tolist() == [float(r + 1) for r in range(world)]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_c4_shape_world_size_4():
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_c4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs == {r: True for r in range(world)}


def test_shard_row_counts():
    assert shard_row_counts(65, 2, 5) == [165, 160] and shard_row_counts(2048, 8, 5) == [1280] * 8


def test_shard_covers_all_streams():
    for n in (1, 5, 64, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_streams(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_make_batch_distinct_layout():
    """synthetic.make_batch(..., distinct=D) -- the layout bench.py's many-rank set-up builds on the device (voxelizer.tile_events):
    streams 0 .. D-1 from their seeds, stream b >= D = stream b % D with coordinates rotated by (7, 3) * (b // D), same timestamps."""
    import numpy as np
    from evfly_amd import synthetic as syn
    H, W, T, B, D = 260, 346, 2, 11, 4
    a = syn.make_batch(D, T, H, W, 500, first_stream=9)
    c = syn.make_batch(B, T, H, W, 500, first_stream=9, distinct=D)
    oa, oc = a["offsets"], c["offsets"]
    assert len(oc) == B + 1 and c["edges"].shape == (B, T + 1) and c["x"].dtype == a["x"].dtype
    for b in range(B):
        k, j = divmod(b, D)
        sa, sc = slice(oa[j], oa[j + 1]), slice(oc[b], oc[b + 1])
        assert np.array_equal(c["t"][sc], a["t"][sa]) and np.array_equal(c["p"][sc], a["p"][sa]) and np.array_equal(c["edges"][b], a["edges"][j])
        assert np.array_equal(c["x"][sc], (a["x"][sa].astype(np.int64) + 7 * k) % W) and np.array_equal(c["y"][sc], (a["y"][sa].astype(np.int64) + 3 * k) % H)
