"""Multi-GPU harness: one process per GPU, streams sharded across ranks, one collective.

The path shards by stream: streams never interact, weights are replicated, and the
only exchange is the final all_gather of the per-rank velocity rows (a few KB per rank over
RCCL/xGMI). The reference has nothing distributed; this exists for the throughput benchmark only.
"""
import torch


def shard_streams(n_streams, rank, world):
    """Contiguous block of streams owned by `rank` (balanced to within one stream)."""
    base, rem = divmod(n_streams, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_row_counts(n_streams, world, rows_per_stream):
    """Velocity rows every rank contributes when `n_streams` streams of `rows_per_stream` steps are sharded with
    `shard_streams`: the same list on every rank, computed without an exchange."""
    return [(s1 - s0) * rows_per_stream for s0, s1 in (shard_streams(n_streams, r, world) for r in range(world))]


def gather_velocities(vel, dist=None, counts=None):
    """all_gather of (rows, 3) velocity tensors in rank order; `dist` = torch.distributed or None.

    `counts` = the per-rank row counts (`shard_row_counts`), identical on every rank: with them the call issues exactly
    one collective and no host synchronisation (the benchmark's per-step call). Without them the counts are exchanged
    by an all_gather of one int64 first -- on EVERY call: nothing is cached, because whether a cache entry exists
    could only be decided from rank-local data (this rank's own row count), and two ranks that decide differently
    issue mismatched collectives."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vel
    world = dist.get_world_size()
    if counts is None:
        rows = torch.tensor([vel.shape[0]], device=vel.device, dtype=torch.int64)
        got = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(got, rows)
        counts = [int(c.item()) for c in got]
    else:
        counts = [int(c) for c in counts]
        if len(counts) != world or counts[dist.get_rank()] != vel.shape[0]:
            raise ValueError(f"gather_velocities: counts {counts} do not describe this call (world {world}, "
                             f"rank {dist.get_rank()} holds {vel.shape[0]} rows)")
    if len(set(counts)) == 1:
        out = torch.empty(world * counts[0], vel.shape[1], device=vel.device, dtype=vel.dtype)
        dist.all_gather_into_tensor(out, vel.contiguous())
        return out
    mx = max(counts)                      # ragged shards: pad to the longest, gather, trim
    pad = torch.zeros(mx, vel.shape[1], device=vel.device, dtype=vel.dtype)
    pad[: vel.shape[0]] = vel
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])
