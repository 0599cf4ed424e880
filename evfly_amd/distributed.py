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


def gather_velocities(vel, dist=None):
    """all_gather of (rows, 3) velocity tensors in rank order; `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vel
    world = dist.get_world_size()
    rows = torch.tensor([vel.shape[0]], device=vel.device, dtype=torch.int64)
    counts = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(counts, rows)
    counts = [int(c.item()) for c in counts]
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
