"""world_size-2 gloo test of the pair/query sharding + all-gather (cotr_amd/dist.py) on CPU.
The model behind the wrapper is the CPU oracle (test infrastructure): what is under test is the
partitioning and the gather, which are device independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cotr_amd.dist import PairShardedModel, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 8, 9, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, B, Q, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    from oracle import cotr_oracle
    sd = synth_state_dict(0)
    img, qs = synth_inputs(B, Q, seed=30)
    calls = []

    def model(samples, queries):
        calls.append(tuple(queries.shape))
        return {'pred_corrs': cotr_oracle.cotr_forward(sd, samples, queries)}

    out = PairShardedModel(model)(img, qs)['pred_corrs']
    torch.save({'out': out, 'calls': calls}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def _worker_fake(rank, world, port, B, Q, out_dir):
    """Same wrapper, a cheap per-pair function as the model (what is under test is the partition + gather, here 2-D)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(6)
    img, qs = torch.randn(B, 3, 4, 4, generator=g), torch.rand(B, Q, 2, generator=g)
    calls = []

    def model(samples, queries):
        calls.append((tuple(samples.shape[:1]), tuple(queries.shape)))
        return {'pred_corrs': queries * 3 - samples.mean(dim=(1, 2, 3))[:, None, None]}

    out = PairShardedModel(model)(img, qs)['pred_corrs']
    torch.save({'out': out, 'want': model(img, qs)['pred_corrs'], 'calls': calls[:-1]}, os.path.join(out_dir, f'f{rank}.pt'))
    dist.destroy_process_group()


def test_pair_x_query_shards_on_four_ranks(tmp_path):
    """2 pairs on 4 ranks (the dense initial pass with 2 patch pairs): two ranks share a pair, each encodes ONLY that pair and
    takes half of its queries; the gathered tensor is the unsharded result, bit for bit."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    B, Q = 2, 9
    mp.spawn(_worker_fake, args=(4, port, B, Q, str(tmp_path)), nprocs=4, join=True)
    res = [torch.load(tmp_path / f'f{r}.pt') for r in range(4)]
    for r in res:
        assert torch.equal(r['out'], res[0]['want'])
    assert [r['calls'] for r in res] == [[((1,), (1, 5, 2))], [((1,), (1, 4, 2))], [((1,), (1, 5, 2))], [((1,), (1, 4, 2))]]


@pytest.mark.parametrize('B,Q', [(3, 5), (1, 7)])
def test_sharded_equals_single_process(tmp_path, B, Q):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, B, Q, str(tmp_path)), nprocs=2, join=True)
    from cotr_amd.utils.synth import synth_state_dict, synth_inputs
    from oracle import cotr_oracle
    img, qs = synth_inputs(B, Q, seed=30)
    torch.set_num_threads(2)
    ref = cotr_oracle.cotr_forward(synth_state_dict(0), img, qs)
    r0, r1 = (torch.load(tmp_path / f'r{r}.pt') for r in range(2))
    assert r0['out'].shape == (B, Q, 2)
    assert torch.equal(r0['out'], r1['out'])                   # every rank holds the full result
    assert float((r0['out'] - ref).abs().max()) < 1e-5          # == unsharded (same CPU arithmetic, different batching)
    if B >= 2:
        assert r0['calls'] == [(2, Q, 2)] and r1['calls'] == [(1, Q, 2)]     # pairs sharded 2 + 1
    else:
        assert r0['calls'] == [(1, 4, 2)] and r1['calls'] == [(1, 3, 2)]     # queries sharded 4 + 3


def _local_shard_worker(rank, world, port, out_dir):
    """Every rank holds only ITS pairs (configs[3] style) and its own numpy RNG state; after the calls all ranks hold
    the full prediction and rank 0's RNG stream."""
    import numpy as np
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cotr_amd.dist import broadcast_numpy_rng, shard_range
    B, Q = 5, 3
    g = torch.Generator().manual_seed(4)
    img, qs = torch.randn(B, 3, 4, 4, generator=g), torch.rand(B, Q, 2, generator=g)

    def model(samples, queries):                                  # any per-pair function
        return {'pred_corrs': queries * 2 + samples.mean(dim=(1, 2, 3))[:, None, None]}

    lo, hi = shard_range(B, world, rank)
    out = PairShardedModel(model, local_shard=True)(img[lo:hi], qs[lo:hi], B=B)['pred_corrs']
    np.random.seed(100 + rank)                                    # ranks disagree ...
    np.random.standard_normal(3)                                  # ... including the cached-gaussian part of the state
    broadcast_numpy_rng()
    draws = np.concatenate([np.random.permutation(10).astype(np.float64), np.random.standard_normal(2)])
    torch.save({'out': out, 'want': model(img, qs)['pred_corrs'], 'draws': draws}, os.path.join(out_dir, f's{rank}.pt'))
    dist.destroy_process_group()


def test_local_shards_and_rng_broadcast(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_local_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f's{r}.pt', weights_only=False) for r in range(2))
    assert torch.equal(r0['out'], r0['want']) and torch.equal(r1['out'], r0['want'])
    import numpy as np
    assert np.array_equal(r0['draws'], r1['draws'])
    np.random.seed(100)
    np.random.standard_normal(3)
    want = np.concatenate([np.random.permutation(10).astype(np.float64), np.random.standard_normal(2)])
    assert np.array_equal(r0['draws'], want)                      # rank 0's stream, unperturbed


# ---- zoom-in tasks sharded over ranks (cotr_amd.dist.sharded_zoom_engine) ------------------------------------------
def _engine_worker(rank, world, port, golden_dir, out_dir):
    import numpy as np
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cotr_amd.dist import sharded_zoom_engine
    from oracle.dense_post import host_dense_post_factory
    from tests.engine_fixtures import CyclicFakeModel, FakeModel, ids, pil_cropper_factory, synthetic_pair
    from tests.test_zoom_engine_cpu import ZOOMS, run_dense_case
    ok = True
    g = np.load(os.path.join(golden_dir, 'engine_c3_filter.npz'))          # known-scale path, converge_iters = 3
    seed, n, conv, force = (int(v) for v in g['meta'])
    img_a, img_b = synthetic_pair(seed)
    eng = sharded_zoom_engine(FakeModel(), max_pairs=16, make_cropper=pil_cropper_factory)
    res = eng.refine(img_a, img_b, g['init'][:, :2], g['init'][:, 2:], 1.0, 1.0, ZOOMS, conv, force=bool(force))
    ok &= np.array_equal(res.loc_history.transpose(1, 0, 2), g['loc_history']) and np.array_equal(res.loc_to, g['best'])
    ok &= res.crops == int(g['total_tasks']) and eng.total_tasks == int(g['total_tasks'])
    for name in ('engine_dense_default_c3', 'engine_cycle_queries'):           # default path incl. early exit
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        eng = sharded_zoom_engine(CyclicFakeModel(), max_pairs=40, make_cropper=pil_cropper_factory,
                                  make_dense_post=host_dense_post_factory)
        out = run_dense_case(g, eng)
        ok &= np.array_equal(out[0], g['corrs']) and np.array_equal(ids(out[1]), g['idx'])
    # the dense initial pass itself is sharded: with 4 patch pairs on 2 ranks each rank's model saw 2 of them (131072 grid
    # queries each), never all 4; with ONE patch pair (square images) the two ranks split its queries
    dense_calls = [c for c in eng.model.calls if c[1][1] == 131072]
    ok &= len(dense_calls) >= 1 and all(c[0][0] == 2 for c in dense_calls)
    sq_a, sq_b = synthetic_pair(3, shape_a=(300, 300), shape_b=(320, 320))
    eng1 = sharded_zoom_engine(CyclicFakeModel(), max_pairs=40, make_cropper=pil_cropper_factory,
                               make_dense_post=host_dense_post_factory)
    solo = sharded_zoom_engine(CyclicFakeModel(), max_pairs=40, make_cropper=pil_cropper_factory,
                               make_dense_post=host_dense_post_factory)
    solo._dense_model = solo.model                                             # the unsharded dense pass, same process
    f1, f0 = eng1.flow(sq_a, sq_b, resample=False), solo.flow(sq_a, sq_b, resample=False)
    ok &= all(np.array_equal(a, b) for a, b in zip(f1, f0) if a is not None)
    ok &= [c[1] for c in eng1.model.calls] == [(1, 65536, 2)] and [c[1] for c in solo.model.calls] == [(1, 131072, 2)]
    torch.save(bool(ok), os.path.join(out_dir, f'e{rank}.pt'))
    dist.destroy_process_group()


def test_sharded_zoom_engine_equals_reference_engine(tmp_path, golden_dir):
    """Two ranks, each refining half of the tasks: every rank ends with the reference engine's golden result."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_engine_worker, args=(2, port, golden_dir, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), f'e{r}.pt')) for r in range(2))


# ---- gradient exchange: chunked GradSink flush with the exchange started behind every chunk (train_ops.GradSink.set_exchange) ----
def _cpu_reduce_jobs(jobs_ptr, srcs_ptr, cmap_ptr, njobs, nchunks):
    """What cotr_train_reduce_jobs does for plain jobs (no scale, no re-layout), on host memory: dst[i] += sum over the job's
    sources (in order), over each source's partials (in order).  Stand-in for the HIP launch in the CPU test of the HOST logic."""
    import ctypes
    import numpy as np
    from cotr_amd import train_ops as T
    job_dt, src_dt = T._record_dtypes()
    jobs = np.frombuffer((ctypes.c_char * (njobs * job_dt.itemsize)).from_address(jobs_ptr), dtype=job_dt)
    nsrc = int(max(j['first_src'] + j['n_src'] for j in jobs))
    srcs = np.frombuffer((ctypes.c_char * (nsrc * src_dt.itemsize)).from_address(srcs_ptr), dtype=src_dt)
    cmap = np.frombuffer((ctypes.c_char * (nchunks * 4)).from_address(cmap_ptr), dtype=np.uint32)
    assert sorted(set(cmap.tolist())) == list(range(njobs))              # every job has its chunks
    for j in jobs:
        n = int(j['numel'])
        assert j['scale'] == 0 and j['taps'] == 1
        dst = np.frombuffer((ctypes.c_char * (n * 4)).from_address(int(j['dst'])), dtype=np.float32)
        acc = dst.copy()
        for sidx in range(int(j['first_src']), int(j['first_src'] + j['n_src'])):
            sr = srcs[sidx]
            for p in range(int(sr['nparts'])):
                part = np.frombuffer((ctypes.c_char * (n * 4)).from_address(int(sr['part']) + p * int(sr['pstride']) * 4), dtype=np.float32)
                acc = acc + part
        dst[:] = acc


def _worker_sink(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cotr_amd import train_ops as T
    from cotr_amd.dist import flat_exchange_async, sync_flat_gradients
    shapes = [(256, 1024), (256,), (1000, 37), (64, 64), (5,), (768, 256), (3, 3)]

    def run(chunked):
        params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
        sink = T.GradSink(params)
        sink._reduce = _cpu_reduce_jobs
        g = torch.Generator().manual_seed(100 + rank)
        keep = []
        for use in range(2):                                            # every parameter is used twice, like a cycle pass
            for p in params:
                nparts = 1 + (p.numel() % 3)
                part = torch.randn(nparts * p.numel(), generator=g)
                keep.append(part)
                sink.add(p.grad.view(-1), part, 0, nparts, p.numel(), p.numel())
        started = []
        if chunked:
            base = flat_exchange_async(None)

            def start(piece):
                started.append((int((piece.data_ptr() - sink.flat.data_ptr()) // 4), piece.numel()))
                return base(piece)
            sink.set_exchange(start, parts=3)
            sink.flush()
            sink.set_exchange(None)
        else:
            sink.flush()
            sync_flat_gradients(sink.flat, None)
        return sink.flat.clone(), started, sink.last_parts, sink.last

    one, _, _, last_one = run(False)
    three, started, parts, last_three = run(True)
    torch.save({'one': one, 'three': three, 'started': started, 'parts': parts, 'last': (last_one, last_three)},
               os.path.join(out_dir, f's{rank}.pt'))
    # an odd world size does not divide the buffer (a multiple of 64 floats): the tail goes through its own tiny all-reduce
    odd = torch.arange(64 * 5 + 0, dtype=torch.float32) * (rank + 1)
    sync_flat_gradients(odd, None)
    torch.save(odd, os.path.join(out_dir, f'odd{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_chunked_sink_flush_with_exchange_equals_flush_then_exchange(tmp_path, world):
    """GradSink.flush() with an exchange installed (training.train_batch on several ranks): the buffer is reduced in three
    address-ordered ranges cut at gradient boundaries, each range's reduce-scatter / all-gather started right behind its reduction;
    the averaged gradients are those of ONE reduction followed by ONE exchange of the whole buffer - bit for bit on two ranks, to
    one rounding of the rank sum's association on three - and identical on every rank.
    World size 3 does not divide the ranges: sync_flat_gradients' tail path."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_sink, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f's{r}.pt') for r in range(world)]
    total = res[0]['one'].numel()
    for r in res:
        assert torch.equal(r['one'], res[0]['one'])                      # every rank holds the same average
        if world == 2:
            assert torch.equal(r['three'], r['one'])                     # chunked = unchunked, bit for bit (a + b is commutative)
        else:
            # three ranks: WHICH rank's partial sum a reduce-scatter element passes through depends on the element's position in
            # the exchanged buffer ((a + b) + c here, (b + c) + a there), and the chunked exchange moves those positions: the same
            # sum over the same ranks in another association - one rounding apart, as between any two collective algorithms
            assert torch.allclose(r['three'], r['one'], rtol=1e-6, atol=2e-6)
        assert float(r['one'].abs().max()) > 0
        assert len(r['parts']) == 3 and r['parts'][0][0] == 0 and r['parts'][-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(r['parts'], r['parts'][1:]))
        assert r['started'] == [(lo, hi - lo) for lo, hi in r['parts']]  # one exchange per range, in address order
        assert r['last'][0] == r['last'][1] == (7, 14)                   # the same 7 jobs / 14 sources either way
    want = sum(torch.arange(64 * 5, dtype=torch.float32) * (r + 1) for r in range(world)) / world
    for r in range(world):
        assert torch.allclose(torch.load(tmp_path / f'odd{r}.pt'), want, rtol=1e-6, atol=0)
