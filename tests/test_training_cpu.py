"""Host-side pieces of the training step (cotr_amd/training.py) that do not need a GPU: gradient averaging over a
world-size-2 gloo group (what runs over RCCL with one rank per GPU), the differentiable lin_sine encoding and the image
position table against the oracle, optimiser groups and checkpoint dictionary of the reference."""
import os
import socket

import numpy as np
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


def test_optimizer_groups_and_checkpoint_format(tmp_path):
    m = build_model(cotr_amd.default_args())
    opt = training.optimizer_for(m, learning_rate=1e-4)
    n_trainable = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert sum(p.numel() for g in opt.param_groups for p in g['params']) == n_trainable      # train_cotr.py:49-53
    assert all(g['lr'] == 1e-4 for g in opt.param_groups)
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
