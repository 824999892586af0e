"""RCCL on the real device with a world of ONE rank: the 1-GPU box cannot run N > 1, but everything that is
size-independent does execute here - the "nccl" (= RCCL on ROCm) communicator initialisation in this process, the
device-tensor collectives cotr_amd/dist.py and bench.py issue (all_gather_into_tensor, broadcast, reduce_scatter_tensor,
all_reduce, barrier) on the HIP model's real outputs, and the sharded engine / gradient-sync code paths with the early
"world == 1" exits disabled.  For N = 8 the same calls run with a larger communicator; their partitioning logic is what
the world-size-2 gloo tests (tests/test_dist_cpu.py) cover."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import cotr_amd
from cotr_amd import dist as cdist
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_inputs, synth_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_world_of_one():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    cdist.FORCE_COLLECTIVES = True
    yield
    cdist.FORCE_COLLECTIVES = False
    dist.destroy_process_group()


def test_pair_sharded_model_through_rccl(rccl_world_of_one):
    assert dist.get_backend() == 'nccl' and cdist.comm_device().type == 'cuda'
    m = build_model(cotr_amd.default_args()).cuda().eval()
    m.load_state_dict(synth_state_dict(0))
    img, qs = synth_inputs(3, 5, seed=40)
    img, qs = img.cuda(), qs.cuda()
    want = m(img, qs)['pred_corrs']
    for wrapped in (cdist.PairShardedModel(m), cdist.PairShardedModel(m, local_shard=True)):
        got = wrapped(img, qs, B=3)['pred_corrs'] if wrapped.local_shard else wrapped(img, qs)['pred_corrs']
        assert got.is_cuda and torch.equal(got, want)              # gathered by RCCL: bit-identical, stays on the device
    # fewer pairs than ranks cannot happen with one rank; the query-sharded branch is the same gather on transposed rows
    finish, work = cdist.all_gather_rows(want.transpose(0, 1).contiguous(), [5], async_op=True)
    assert torch.equal(finish().transpose(0, 1), want)
    # bench.py's gather of one step's predictions, asynchronous w.r.t. the next forward
    gathered = torch.empty_like(want)
    w = dist.all_gather_into_tensor(gathered, want, async_op=True)
    again = m(img, qs)['pred_corrs']
    w.wait()
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(gathered, want) and torch.equal(again, want)


def test_rng_broadcast_and_gradient_sync_through_rccl(rccl_world_of_one):
    np.random.seed(5)
    np.random.standard_normal(1)
    want = np.random.get_state()
    cdist.broadcast_numpy_rng()                                   # device broadcast of the packed state, round trip
    got = np.random.get_state()
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[2:] == want[2:]
    params = [torch.nn.Parameter(torch.zeros(s, device='cuda')) for s in ((3, 5), (1000,), (17, 2))]
    g = torch.Generator().manual_seed(1)
    grads = [torch.randn(p.shape, generator=g).cuda() for p in params]
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    cdist.sync_gradients_sharded(params, bucket_elems=1004)       # reduce_scatter_tensor + all_gather_into_tensor on RCCL
    assert all(torch.equal(p.grad, gr) for p, gr in zip(params, grads))       # mean over one rank


def test_sharded_zoom_engine_through_rccl(rccl_world_of_one, golden_dir):
    """The packed float64 result gather of the sharded zoom engine on device buffers; golden of the reference engine."""
    from tests.engine_fixtures import FakeModel, synthetic_pair
    g = np.load(os.path.join(golden_dir, 'engine_c3_filter.npz'))
    seed, n, conv, force = (int(v) for v in g['meta'])
    img_a, img_b = synthetic_pair(seed)
    eng = cdist.sharded_zoom_engine(FakeModel().cuda(), max_pairs=16)        # device crops
    res = eng.refine(img_a, img_b, g['init'][:, :2], g['init'][:, 2:], 1.0, 1.0, np.linspace(0.5, 0.0625, 4), conv,
                     force=bool(force))
    assert np.array_equal(res.loc_history.transpose(1, 0, 2), g['loc_history']) and np.array_equal(res.loc_to, g['best'])
    assert res.crops == int(g['total_tasks'])
