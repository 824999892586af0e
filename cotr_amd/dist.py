"""Multi-GPU: one process per MI355X, pairs (or, when there are fewer pairs than GPUs, queries)
sharded across ranks, no collective in the math, one all-gather of the predicted (x, y).

Pairs never interact anywhere in COTR and queries of a pair do not interact either (the decoder has
no query self-attention, COTR/models/transformer.py:185-201), so the sharded result is bit-identical
to the single-GPU one.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs
of a node); the message is B*Q*2 floats in total - latency-bound, so one padded all_gather is enough.
The same code runs on the "gloo" backend with CPU tensors (tests/test_dist_cpu.py).
"""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous block of ``n`` units for ``rank``; the remainder goes to the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local, counts, group=None, async_op=False):
    """Concatenate per-rank row blocks [n_r, ...] (n_r = counts[r]) along dim 0 on every rank.
    Returns (finish, work): ``finish()`` waits (if async) and returns the gathered tensor."""
    world = dist.get_world_size(group)
    pad = max(counts)
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != pad:
        send = torch.zeros((pad,) + tail, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * pad,) + tail, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        parts = [recv[r * pad: r * pad + counts[r]] for r in range(world)]
        return torch.cat(parts, dim=0)

    return finish, work


class PairShardedModel:
    """Wraps any ``model(samples, queries) -> {'pred_corrs'}``.  Every rank passes the SAME full
    batch; each computes its shard and all ranks return the full [B,Q,2] prediction."""

    def __init__(self, model, group=None):
        self.model = model
        self.group = group

    def __call__(self, samples, queries):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return self.model(samples, queries)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B, Q = queries.shape[:2]
        if B >= world:  # shard image pairs
            lo, hi = shard_range(B, world, rank)
            if hi > lo:
                local = self.model(samples[lo:hi], queries[lo:hi])['pred_corrs']
            else:
                local = queries.new_zeros((0, Q, 2))
            counts = [shard_range(B, world, r)[1] - shard_range(B, world, r)[0] for r in range(world)]
            finish, _ = all_gather_rows(local, counts, self.group)
            return {'pred_corrs': finish()}
        # fewer pairs than ranks: every rank encodes all pairs, decodes a slice of the queries
        lo, hi = shard_range(Q, world, rank)
        if hi > lo:
            local = self.model(samples, queries[:, lo:hi])['pred_corrs']
        else:
            local = queries.new_zeros((B, 0, 2))
        counts = [shard_range(Q, world, r)[1] - shard_range(Q, world, r)[0] for r in range(world)]
        finish, _ = all_gather_rows(local.transpose(0, 1).contiguous(), counts, self.group)  # rows = queries
        return {'pred_corrs': finish().transpose(0, 1).contiguous()}


def sharded_zoom_engine(*args, group=None, **kwargs):
    """``ZoomEngine`` whose zoom-in refinement is sharded over the ranks of ``group``: tasks (one query each) are
    independent (COTR/inference/refinement_task.py: a task only sees its own crops), so every rank refines a contiguous
    block with its own GPU and the per-task results (a few floats each) are all-gathered; no collective in the data path.
    The dense initial pass, task generation and the early-exit bookkeeping are replicated (deterministic, same RNG
    state on every rank), so all ranks return the same correspondences as a single-GPU run, bit for bit."""
    from .inference.zoom_engine import RefineResult, ZoomEngine
    import numpy as np

    class ShardedZoomEngine(ZoomEngine):
        def refine(self, img_a, img_b, loc_from, loc_to, *a, **kw):
            if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
                return super().refine(img_a, img_b, loc_from, loc_to, *a, **kw)
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            loc_from = np.array(loc_from, dtype=np.float64).reshape(-1, 2)
            loc_to = np.array(loc_to, dtype=np.float64).reshape(-1, 2)
            lo, hi = shard_range(len(loc_from), world, rank)
            local = super().refine(img_a, img_b, loc_from[lo:hi], loc_to[lo:hi], *a, **kw)
            parts = [None] * world
            dist.all_gather_object(parts, (local.loc_to, local.good, local.loc_history, local.steps, local.model_calls,
                                           local.crops), group=group)
            # every rank also counts the crops of the other ranks: total_tasks stays the whole job's number
            self.total_tasks += sum(p[5] for r, p in enumerate(parts) if r != rank)
            # RefineResult(loc_from, loc_to, good, loc_history [levels+1, N, 2], model_calls, crops, steps)
            return RefineResult(loc_from, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                                np.concatenate([p[2] for p in parts], axis=1), max(p[4] for p in parts),
                                sum(p[5] for p in parts), np.concatenate([p[3] for p in parts]))

    return ShardedZoomEngine(*args, **kwargs)
