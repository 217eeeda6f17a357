"""Multi-GPU driver pieces: one process per GPU, sequences sharded across ranks, one collective.

The reference has no distributed code at all; what shards naturally is the SEQUENCE
("video"): frame t+1 of a sequence consumes frame t's poses, clusters and MLP weights
(mlp_reg.py:293-378), so frames inside a sequence stay on one GPU.  Ranks therefore own disjoint
sequences and exchange nothing until the end, when the per-frame cluster poses are gathered with
ONE all_gather (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The payload is
frames x K x 64 B (64 KB for 50 x 20): latency-bound, so a single fused gather, not a ring of
small messages.
"""
import torch


def shard_sequences(n_sequences: int, rank: int, world: int):
    """Round-robin ownership: rank r registers sequences r, r + world, ...  (SURVEY.md 8e)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_sequences, world))


def gather_poses(local: torch.Tensor, counts=None, group=None) -> torch.Tensor:
    """local: (F_r, K, 4, 4) poses registered by this rank (F_r may differ between ranks when
    `counts`, the per-rank frame counts, is given).  Returns (sum F_r, K, 4, 4) on every rank, rank-major.
    Single process (no initialised process group): returns `local` unchanged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if counts is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    fmax = max(counts)
    pad = torch.zeros((fmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * fmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.reshape((world, fmax) + tuple(local.shape[1:]))
    return torch.cat([out[r, : counts[r]] for r in range(world)], 0)


def scatter_order(n_sequences: int, world: int, frames_per_seq: int):
    """Index map from the rank-major gathered layout back to (sequence, frame) order."""
    order = []
    for r in range(world):
        for s in shard_sequences(n_sequences, r, world):
            order.extend((s, f) for f in range(frames_per_seq))
    return order
