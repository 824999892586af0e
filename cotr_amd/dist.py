"""Multi-GPU: one process per MI355X, pairs (or, when there are fewer pairs than GPUs, queries)
sharded across ranks, no collective in the math, one all-gather of the predicted (x, y).

Pairs never interact anywhere in COTR and queries of a pair do not interact either (the decoder has
no query self-attention, COTR/models/transformer.py:185-201), so the sharded result is bit-identical
to the single-GPU one.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs
of a node); the message is B*Q*2 floats in total - latency-bound, so one padded all_gather is enough.
The same code runs on the "gloo" backend with CPU tensors (tests/test_dist_cpu.py) and on a
world-size-1 "nccl" group on the single-GPU box (tests/test_dist_gpu.py: RCCL initialisation and the
device-tensor collectives really execute there).

Every exchange in this module is a TENSOR collective (all_gather_into_tensor / broadcast /
reduce_scatter_tensor): nothing is pickled, so the same calls are valid on RCCL device buffers.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous block of ``n`` units for ``rank``; the remainder goes to the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# A world of ONE rank needs no exchange and the helpers below return early.  The single-GPU box has no second rank to talk
# to, so tests/test_dist_gpu.py sets this to True: the collectives are then issued even at world size 1 and really go through
# RCCL on device buffers (communicator initialisation, all_gather_into_tensor, broadcast, reduce_scatter_tensor).
FORCE_COLLECTIVES = False


def _active(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def comm_device(group=None):
    """Where tensors handed to the collectives of ``group`` must live: the current GPU for RCCL, the host for gloo."""
    if dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


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
    """Wraps any ``model(samples, queries) -> {'pred_corrs'}``; all ranks return the full [B,Q,2] prediction.

    ``local_shard=False`` (default): every rank passes the SAME full batch and computes its block of it (pairs; or
    queries when there are fewer pairs than ranks).
    ``local_shard=True``: every rank passes ONLY its own block of pairs - rank r holds pairs shard_range(B, world, r) of a
    global batch of ``B`` pairs (BASELINE.json configs[3]: 256 pairs = 32 per GPU, i.e. 50 MB of images per rank instead
    of 402 MB) - and ``B`` is given to the call.  The gathered prediction is the same."""

    def __init__(self, model, group=None, local_shard=False):
        self.model = model
        self.group = group
        self.local_shard = local_shard

    def __call__(self, samples, queries, B=None):
        if not _active(self.group):
            return self.model(samples, queries)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if self.local_shard:
            assert B is not None, 'local_shard=True: pass the global number of pairs B'
            Q = queries.shape[1]
            lo, hi = shard_range(B, world, rank)
            assert samples.shape[0] == hi - lo == queries.shape[0], (samples.shape[0], lo, hi)
            local = self.model(samples, queries)['pred_corrs'] if hi > lo else queries.new_zeros((0, Q, 2))
            counts = [shard_range(B, world, r)[1] - shard_range(B, world, r)[0] for r in range(world)]
            finish, _ = all_gather_rows(local, counts, self.group)
            return {'pred_corrs': finish()}
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
        if B > 0 and world % B == 0:
            # fewer pairs than ranks, ranks a multiple of the pairs (the dense initial pass: 1, 2 or 4 patch pairs x 131072 grid
            # queries on 8 GPUs): world / B ranks share one pair - each encodes ONLY that pair and decodes its slice of the
            # pair's queries (queries are independent: no query self-attention, transformer.py:185-201).  Rank order = (pair,
            # slice), so the gathered rows are the [B, Q, 2] tensor as it lies in memory.
            g = world // B
            pair, sub = rank // g, rank % g
            lo, hi = shard_range(Q, g, sub)
            if hi > lo:
                local = self.model(samples[pair:pair + 1], queries[pair:pair + 1, lo:hi])['pred_corrs'][0]
            else:
                local = queries.new_zeros((0, 2))
            counts = [shard_range(Q, g, r % g)[1] - shard_range(Q, g, r % g)[0] for r in range(world)]
            finish, _ = all_gather_rows(local.contiguous(), counts, self.group)
            return {'pred_corrs': finish().view(B, Q, 2)}
        # otherwise: every rank encodes all pairs, decodes a slice of the queries
        lo, hi = shard_range(Q, world, rank)
        if hi > lo:
            local = self.model(samples, queries[:, lo:hi])['pred_corrs']
        else:
            local = queries.new_zeros((B, 0, 2))
        counts = [shard_range(Q, world, r)[1] - shard_range(Q, world, r)[0] for r in range(world)]
        finish, _ = all_gather_rows(local.transpose(0, 1).contiguous(), counts, self.group)  # rows = queries
        return {'pred_corrs': finish().transpose(0, 1).contiguous()}


def broadcast_numpy_rng(group=None, src=0):
    """Make numpy's GLOBAL RNG state on every rank equal to rank ``src``'s (one 632-element int64 tensor broadcast).
    The engines draw from it where the reference does (np.random.choice in gen_tasks, np.random.permutation in
    FasterSparseEngine): replicated host logic must see replicated draws, whatever each rank did before."""
    if not _active(group):
        return
    dev = comm_device(group)
    name, keys, pos, has_gauss, cached = np.random.get_state()
    assert name == 'MT19937'
    buf = torch.zeros(632, dtype=torch.int64)
    buf[:624] = torch.from_numpy(keys.astype(np.int64))
    buf[624], buf[625] = int(pos), int(has_gauss)
    buf[626] = int(np.float64(cached).view(np.int64))
    buf = buf.to(dev)
    dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    buf = buf.cpu().numpy()
    np.random.set_state(('MT19937', buf[:624].astype(np.uint32), int(buf[624]), int(buf[625]),
                         float(np.int64(buf[626]).view(np.float64))))


def sharded_zoom_engine(*args, group=None, **kwargs):
    """``ZoomEngine`` whose zoom-in refinement is sharded over the ranks of ``group``: tasks (one query each) are
    independent (COTR/inference/refinement_task.py: a task only sees its own crops), so every rank refines a contiguous
    block with its own GPU and the per-task results are gathered as ONE packed float64 tensor
    (all_gather_into_tensor; [tasks, 4 + 2*(levels+1)] + one bookkeeping row per rank); no collective in the data path.
    The dense initial pass (inference_helper.py:105-165: every patch pair x the 131072-query grid - most of config 2's
    model time before the zoom levels) is sharded too: its one model call goes through ``PairShardedModel`` (patch pairs over
    the ranks; with fewer pairs than ranks the ranks of a pair split its queries and encode only that pair), ONE all-gather of
    the [P,131072,2] prediction, then the post-processing, task generation and early-exit bookkeeping run replicated on the
    gathered tensor; numpy's global RNG, which task generation draws from, is synchronised from rank 0 first.  All ranks
    return the same correspondences; they equal a single-GPU run with rank 0's RNG state bit for bit when the model gives the
    same bits for a pair whatever the batch shape (the fake models of the tests; the HIP model may pick another GEMM
    configuration for another shape and then agrees to ~1e-4 px)."""
    from .inference.zoom_engine import RefineResult, ZoomEngine

    class ShardedZoomEngine(ZoomEngine):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self._dense_model = PairShardedModel(self.model, group)     # ZoomEngine.flow calls this for the dense pass

        def gen_tasks(self, *a, **kw):
            broadcast_numpy_rng(group)
            return super().gen_tasks(*a, **kw)

        def refine(self, img_a, img_b, loc_from, loc_to, *a, **kw):
            if not _active(group):
                return super().refine(img_a, img_b, loc_from, loc_to, *a, **kw)
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            loc_from = np.array(loc_from, dtype=np.float64).reshape(-1, 2)
            loc_to = np.array(loc_to, dtype=np.float64).reshape(-1, 2)
            n = len(loc_from)
            lo, hi = shard_range(n, world, rank)
            local = super().refine(img_a, img_b, loc_from[lo:hi], loc_to[lo:hi], *a, **kw)
            lv = local.loc_history.shape[0]                                  # levels + 1
            width = 4 + 2 * lv
            rows = np.zeros((hi - lo + 1, width), dtype=np.float64)          # last row: this rank's bookkeeping
            rows[:-1, 0:2] = local.loc_to
            rows[:-1, 2] = local.good
            rows[:-1, 3] = local.steps
            rows[:-1, 4:] = local.loc_history.transpose(1, 0, 2).reshape(hi - lo, 2 * lv)
            rows[-1, 0], rows[-1, 1] = local.model_calls, local.crops
            counts = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] + 1 for r in range(world)]
            finish, _ = all_gather_rows(torch.from_numpy(rows).to(comm_device(group)), counts, group)
            allrows = finish().cpu().numpy()
            book = np.cumsum(counts) - 1                                     # index of every rank's bookkeeping row
            tasks = np.delete(allrows, book, axis=0)
            calls, crops = allrows[book, 0], allrows[book, 1]
            # every rank also counts the crops of the other ranks: total_tasks stays the whole job's number
            self.total_tasks += int(crops.sum() - crops[rank])
            hist = tasks[:, 4:].reshape(n, lv, 2).transpose(1, 0, 2)
            # RefineResult(loc_from, loc_to, good, loc_history [levels+1, N, 2], model_calls, crops, steps, last_iters);
            # the per-iteration positions of the last level stay on the rank that computed them
            return RefineResult(loc_from, tasks[:, 0:2].copy(), tasks[:, 2] > 0.5, np.ascontiguousarray(hist),
                                int(calls.max()), int(crops.sum()), tasks[:, 3].astype(np.int64), None)

    return ShardedZoomEngine(*args, **kwargs)


def sync_flat_gradients(flat, group=None):
    """The same exchange for gradients that already ARE one flat buffer (``train_ops.GradSink.flat``): reduce-scatter, 1/N on the
    local shard, all-gather in place - no packing copies.  The buffer's length is a multiple of 64 floats, which a world size of
    3, 5, 6 or 7 does not divide: the largest multiple of the world size goes through the two collectives, the (< world) elements
    behind it through one tiny all-reduce."""
    if not _active(group):
        return
    world = dist.get_world_size(group)
    main = flat.numel() // world * world
    if main:
        head = flat[:main]
        shard = torch.empty(main // world, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(shard, head, op=dist.ReduceOp.SUM, group=group)
        shard /= world
        dist.all_gather_into_tensor(head, shard, group=group)
    if main < flat.numel():
        tail = flat[main:]
        dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group)
        tail /= world


def flat_exchange_async(group=None):
    """-> ``start(flat_slice) -> finish()`` for ``train_ops.GradSink.set_exchange``: the reduce-scatter / (1/N) / all-gather
    exchange of ``sync_flat_gradients`` on a contiguous slice of the flat gradient buffer, issued on a side stream so that the
    caller's stream goes on (to the reduction launch of the next slice) while the collectives run; ``finish()`` makes the caller's
    stream wait for them.  On CPU tensors (gloo tests) the exchange simply runs in ``start``.  Results per element are those of
    one exchange over the whole buffer: a sum over the same ranks."""
    def start(piece):
        if not _active(group):
            return lambda: None
        if not piece.is_cuda:
            sync_flat_gradients(piece, group)
            return lambda: None
        cur = torch.cuda.current_stream(piece.device)
        side = _side_stream(piece.device)
        side.wait_stream(cur)                                   # the slice's reduction launch is enqueued on `cur`
        with torch.cuda.stream(side):
            sync_flat_gradients(piece, group)
        return lambda: cur.wait_stream(side)
    return start


_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def sync_gradients_sharded(params, group=None, bucket_elems=1 << 25):
    """Average the gradients over the ranks: per flat fp32 bucket (128 MB: the whole trainable part - 39.5 MB in stage 1
    of the reference's recipe, 72.7 MB with the backbone, SURVEY.md 8f4 - is ONE bucket) one reduce-scatter + one
    all-gather (padded to a multiple of the world size).  On a fully connected xGMI node every GPU then moves (N-1)/N of
    the buffer over its 7 links in parallel, twice, instead of walking a ring, and the 1/N average is applied to the
    local shard between the two collectives (N times fewer multiplies).  Same result as an all-reduce up to fp32
    summation order; few large messages because xGMI links are per-link bound."""
    if not _active(group):
        return
    world = dist.get_world_size(group)
    params = [p for p in params if p.grad is not None]
    i = 0
    while i < len(params):
        j, total = i, 0
        while j < len(params) and (total == 0 or total + params[j].numel() <= bucket_elems):
            total += params[j].numel()
            j += 1
        padded = (total + world - 1) // world * world
        flat = torch.zeros(padded, dtype=torch.float32, device=params[i].grad.device)
        off = 0
        for p in params[i:j]:
            flat[off:off + p.numel()] = p.grad.reshape(-1)
            off += p.numel()
        shard = torch.empty(padded // world, dtype=torch.float32, device=flat.device)
        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
        shard /= world
        dist.all_gather_into_tensor(flat, shard, group=group)
        off = 0
        for p in params[i:j]:
            p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()
        i = j
