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


_COUNTS = {}      # (world, rank-local rows) -> per-rank row counts, exchanged once


def gather_velocities(vel, dist=None):
    """all_gather of (rows, 3) velocity tensors in rank order; `dist` = torch.distributed or None.
    The per-rank row counts are exchanged on the first call for a given local row count and cached (the
    benchmark calls this once per step with fixed shapes: no count exchange or host sync inside the timed loop).
    Contract: every rank keeps its own row count fixed between calls that share a cache entry."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vel
    world = dist.get_world_size()
    key = (world, vel.shape[0])
    counts = _COUNTS.get(key)
    if counts is None:
        rows = torch.tensor([vel.shape[0]], device=vel.device, dtype=torch.int64)
        got = [torch.zeros_like(rows) for _ in range(world)]
        dist.all_gather(got, rows)
        counts = _COUNTS[key] = [int(c.item()) for c in got]
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
