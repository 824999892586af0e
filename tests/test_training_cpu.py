"""Host-side pieces of the training step (cotr_amd/training.py) that do not need a GPU: gradient averaging over a
world-size-2 gloo group (what runs over RCCL with one rank per GPU), the differentiable lin_sine encoding and the image
position table against the oracle, optimiser groups and checkpoint dictionary of the reference."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from oracle import cotr_oracle


def test_encodings_match_oracle():
    x = torch.rand(7, 5, 2)
    assert torch.equal(training.lin_sine(x), cotr_oracle.nerf_positional_encoding(x))
    pos = cotr_oracle.image_position_embedding(1, 16, 32, torch.float32)[0]            # [256,16,32]
    assert torch.equal(training.image_pos_table('cpu'), pos.permute(1, 2, 0).reshape(512, 256))
    x = torch.rand(4, 2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: training.lin_sine(t, depth=3), (x,))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((3, 5), (1000,), (17, 2), (4,))]
    grads = [torch.randn(world, *p.shape, generator=g) for p in params]               # same on every rank
    for p, gr in zip(params, grads):
        p.grad = gr[rank].clone()
    params.append(torch.nn.Parameter(torch.zeros(2)))                                  # no gradient: skipped
    training.sync_gradients(params, bucket_elems=1004)                                # forces several buckets
    ok = all(torch.allclose(p.grad, gr.mean(0), atol=1e-7) for p, gr in zip(params, grads)) and params[-1].grad is None
    torch.save(ok, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_gradient_sync_world_size_2(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(2))


def _nan_worker(rank, world, port, out_dir):
    """rank 0 sees a NaN loss, rank 1 a finite one: both must skip the step together (no collective mismatch)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(3)                                  # identical replicas
    lin = torch.nn.Linear(4, 2)
    lin.training = True
    before = [p.detach().clone() for p in lin.parameters()]
    opt = torch.optim.Adam(lin.parameters(), lr=0.1)
    x = torch.ones(3, 4)
    results = []
    for step, nan_rank in enumerate((0, None)):            # step 0: rank 0 is NaN -> nobody steps; step 1: everybody steps
        def fake_loss(model, img, query, target, *a):
            out = model(x)
            loss = out.sum() * (float('nan') if rank == nan_rank else 1.0 + rank)
            return loss, out
        training.compute_loss, saved = fake_loss, training.compute_loss
        try:
            value, _ = training.train_batch(lin, opt, None, None, None)
        finally:
            training.compute_loss = saved
        results.append([p.detach().clone() for p in lin.parameters()])
    unchanged = all(torch.equal(a, b) for a, b in zip(before, results[0]))
    moved = not all(torch.equal(a, b) for a, b in zip(results[0], results[1]))
    torch.save((unchanged, moved, results[1]), os.path.join(out_dir, f'n{rank}.pt'))
    dist.destroy_process_group()


def test_nan_loss_is_skipped_by_all_ranks_together(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_nan_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f'n{r}.pt')) for r in range(2))
    assert r0[0] and r1[0] and r0[1] and r1[1]
    assert all(torch.allclose(a, b) for a, b in zip(r0[2], r1[2]))          # averaged gradients: replicas stay identical


def test_optimizer_groups_and_checkpoint_format(tmp_path):
    m = build_model(cotr_amd.default_args())
    opt = training.optimizer_for(m, learning_rate=1e-4)
    n_trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert sum(p.numel() for g in opt.param_groups for p in g['params']) == n_trainable      # train_cotr.py:49-53
    assert all(g['lr'] == 1e-4 for g in opt.param_groups)
    # train_cotr.py:49-53 builds FOUR groups (the query_proj one is empty): a reference optim_state_dict must load
    assert [len(g['params']) > 0 for g in opt.param_groups] == [True, True, False, True]
    ref_like = torch.optim.Adam([{'params': list(m.transformer.parameters()), 'lr': 1e-4},
                                 {'params': list(m.corr_embed.parameters()), 'lr': 1e-4},
                                 {'params': list(m.query_proj.parameters()), 'lr': 1e-4},
                                 {'params': list(m.input_proj.parameters()), 'lr': 1e-4}])
    opt.load_state_dict(ref_like.state_dict())
    assert len(training.optimizer_for(m, 1e-4, lr_backbone=1e-5).param_groups) == 5                  # :54-55
    path = os.path.join(str(tmp_path), 'checkpoint.pth.tar')
    training.save_checkpoint(path, m, opt, epoch=3, iteration=1234)
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'epoch', 'iteration', 'optim_state_dict', 'model_state_dict'}          # cotr_trainer.py:75-81
    assert set(ck['model_state_dict']) == set(m.state_dict())
    m2 = build_model(cotr_amd.default_args())
    assert training.load_checkpoint(path, m2) == (3, 1234)
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())


def test_training_mode_without_gpu_fails_loudly():
    m = build_model(cotr_amd.default_args()).train()
    try:
        m(torch.zeros(1, 3, 256, 512), torch.zeros(1, 4, 2))
    except Exception as e:                                          # no CPU fallback, in training mode either
        assert 'MI355X' in str(e) or 'HIP' in str(e) or 'cuda' in str(e).lower(), e
    else:
        raise AssertionError('training forward on CPU tensors must raise')


def test_grad_sink_tables_and_ownership():
    """train_ops.GradSink, host side: flat gradient views, ownership of parameters and of same-size views of them, the job /
    source records handed to cotr_train_reduce_jobs (grouping of several uses of one parameter, chunk counts, the vector flag)."""
    from cotr_amd import train_ops as T
    w = torch.nn.Parameter(torch.zeros(256, 1024, 1, 1))
    b = torch.nn.Parameter(torch.zeros(256))
    other = torch.nn.Parameter(torch.zeros(8))
    frozen = torch.nn.Parameter(torch.zeros(8), requires_grad=False)
    sink = T.GradSink([w, b, frozen])
    assert sink.flat.numel() == 256 * 1024 + 256 and w.grad.shape == w.shape and b.grad.data_ptr() == sink.flat.data_ptr() + 256 * 1024 * 4
    assert sink.grad_of(w) is not None and sink.grad_of(other) is None and sink.grad_of(frozen) is None
    view = sink.grad_of(w.view(256, 1024))                # input_proj.weight.view(d, CFEAT) in forward_train
    assert view is not None and view.shape == (256, 1024) and view.data_ptr() == w.grad.data_ptr()
    assert sink.grad_of(w.view(256, 1024)[:128]) is None and sink.grad_of(w.detach()) is None
    part1, part2 = torch.zeros(3 * (512 * 1024 + 512) + 4), torch.zeros(2 * (512 * 1024 + 512))
    rec = 512 * 1024 + 512
    sink.add(view[0:128], part1, 0, 3, rec, 128 * 1024)
    sink.add(b.grad[0:128], part1, 512 * 1024, 3, rec, 128)
    sink.add(view[128:256], part1[4:], 0, 3, rec, 128 * 1024)     # (16-byte aligned all the same)
    sink.add(view[0:128], part2, 0, 2, rec, 128 * 1024)           # a second use of the first slice
    sink.add(b.grad[128:130], part2, 2, 2, rec, 2)                # two floats, source 8 bytes off: scalar path
    jobs, srcs, nchunks = sink.tables()
    assert len(jobs) == 4 and len(srcs) == 5 and nchunks == 128 + 1 + 128 + 1
    # jobs ordered by the number of partials a thread walks (5, 3, 3, 2), sources of a job in registration order
    assert [int(j['dst']) for j in jobs] == [view[0:128].data_ptr(), b.grad.data_ptr(), view[128:256].data_ptr(), b.grad[128:130].data_ptr()]
    assert list(jobs['n_src']) == [2, 1, 1, 1] and list(jobs['first_src']) == [0, 2, 3, 4]
    assert list(jobs['chunk0']) == [0, 128, 129, 257] and list(jobs['vec']) == [1, 1, 1, 0]
    assert srcs['part'][1] == part2.data_ptr() and srcs['nparts'][1] == 2 and srcs['pstride'][0] == rec
    cmap = T.GradSink.chunk_map(jobs, nchunks)
    assert len(cmap) == 258 and cmap[0] == 0 and cmap[127] == 0 and cmap[128] == 1 and cmap[129] == 2 and cmap[257] == 3
    with pytest.raises(AssertionError):
        sink.add(other, part1, 0, 1, 8, 8)                        # not a view into this sink
    sink.discard()
    assert sink.tables()[2] == 0
    w.grad = None
    sink.attach()                                                 # after optim.zero_grad(set_to_none=True)
    assert w.grad is not None and w.grad.data_ptr() == sink.flat.data_ptr()


def _sink_worker(rank, world, port, out_dir):
    """train_batch(..., sink=...) over two gloo ranks: the flat buffer is averaged by ONE reduce-scatter + all-gather."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(3)
    lin = torch.nn.Linear(4, 2)
    lin.training = True
    opt = torch.optim.SGD(lin.parameters(), lr=1.0)
    sink = training.grad_sink_for(opt)
    x = torch.ones(3, 4)

    def fake_loss(model, img, query, target, *a):
        out = model(x)
        return out.sum() * (1.0 + rank), out
    before = [p.detach().clone() for p in lin.parameters()]
    training.compute_loss, saved = fake_loss, training.compute_loss
    try:
        for _ in range(2):                                       # two steps: the buffer is zeroed in between
            training.train_batch(lin, opt, None, None, None, sink=sink)
    finally:
        training.compute_loss = saved
    # d(sum)/dW = 3 per element, db = 3; x (1 + rank) averaged over ranks 0, 1 = x 1.5; two SGD steps of lr 1
    ok = all(torch.allclose(b - 2 * 4.5, a) for a, b in zip(lin.parameters(), before))
    torch.save((ok, [p.detach().clone() for p in lin.parameters()]), os.path.join(out_dir, f's{rank}.pt'))
    dist.destroy_process_group()


def test_train_batch_with_sink_world_size_2(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_sink_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f's{r}.pt')) for r in range(2)]
    assert res[0][0] and res[1][0]
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
