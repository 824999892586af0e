"""ZoomEngine on the MI355X: (1) with the HIP crop+resize kernel and the fake model it reproduces the reference
engine's golden trajectories bit for bit (the device crops are Pillow-exact); (2) with the real HIP model one zoom
level agrees with the CPU oracle on the same crops to the parity bar."""
import os

import numpy as np
import pytest
import torch

import cotr_amd
from cotr_amd.inference import ZoomEngine
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
from oracle import cotr_oracle, dense_post
from tests.engine_fixtures import FakeModel, ids, synthetic_pair, pil_cropper_factory

pytestmark = pytest.mark.gpu
ZOOMS = np.linspace(0.5, 0.0625, 4)


@pytest.mark.parametrize('name', ['engine_c1_force', 'engine_c3_force', 'engine_c3_filter'])
def test_device_crops_reproduce_reference_trajectories(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    seed, n, conv, force = (int(v) for v in g['meta'])
    img_a, img_b = synthetic_pair(seed)
    model = FakeModel().cuda()                       # parameters on the GPU -> the engine crops on the device
    eng = ZoomEngine(model, max_pairs=64)
    res = eng.refine(img_a, img_b, g['init'][:, :2], g['init'][:, 2:], 1.0, 1.0, ZOOMS, conv, force=bool(force))
    assert np.array_equal(res.loc_history.transpose(1, 0, 2), g['loc_history'])
    assert np.array_equal(res.loc_to, g['best'])


class _OracleModel(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.sd = sd
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, img, q):
        return {'pred_corrs': cotr_oracle.cotr_forward(self.sd, img, q)}


def test_one_zoom_level_with_the_real_model_matches_oracle():
    sd = synth_state_dict(0)
    img_a, img_b = synthetic_pair(5)
    rng = np.random.default_rng(1)
    n = 12
    loc_from = np.stack([rng.uniform(5, img_a.shape[1] - 5, n), rng.uniform(5, img_a.shape[0] - 5, n)], 1)
    loc_to = np.stack([rng.uniform(5, img_b.shape[1] - 5, n), rng.uniform(5, img_b.shape[0] - 5, n)], 1)
    hip = build_model(cotr_amd.default_args()).cuda().eval()
    hip.load_state_dict(sd)
    got = ZoomEngine(hip).refine(img_a, img_b, loc_from, loc_to, 1.0, 1.0, [0.5], 1, force=True)
    ref = ZoomEngine(_OracleModel(sd), make_cropper=pil_cropper_factory).refine(img_a, img_b, loc_from, loc_to, 1.0, 1.0,
                                                                                [0.5], 1, force=True)
    size_b = int((min(img_b.shape[:2]) * 0.5 // 2) * 2)
    # 1e-3 px in the 256x512 network frame = 1e-3 * size/256 px in the image
    assert np.abs(got.loc_to - ref.loc_to).max() < 1e-3 * size_b / 256 * 2
    assert got.crops == n and got.model_calls == 1          # one launch for the whole level


def test_corr_base_reuses_one_encode_for_the_cycle_pass():
    """model.encode + 2x model.decode (K/V cached in the handle) == two full forwards (inference_helper.py:197-198)."""
    sd = synth_state_dict(0)
    img_a, img_b = synthetic_pair(6)                    # non-square: 2 x 2 patch pairs
    rng = np.random.default_rng(2)
    q = np.stack([rng.uniform(5, img_a.shape[1] - 5, 30), rng.uniform(5, img_a.shape[0] - 5, 30)], 1)
    hip = build_model(cotr_amd.default_args()).cuda().eval()
    hip.load_state_dict(sd)

    class NoSplit(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, img, qs):
            return self.m(img, qs)

    split = ZoomEngine(hip).corr_base(img_a, img_b, q)
    full = ZoomEngine(NoSplit(hip)).corr_base(img_a, img_b, q)
    assert split.shape == (30, 4) and np.array_equal(split, full)


# ---- default path (dense initial pass, task generation, early exit, cycle-consistency wrapper) -----------------------
DENSE_CASES = ['engine_dense_default', 'engine_dense_default_c3', 'engine_dense_queries_filter',
               'engine_dense_queries_force', 'engine_cycle_default', 'engine_cycle_queries']


@pytest.mark.parametrize('name', DENSE_CASES)
def test_default_path_on_the_device_equals_the_reference_engine(name, golden_dir):
    """The DEFAULT ZoomEngine - crops cut by the HIP kernel AND the dense pass post-processed on the device (cotr_dense_cycle
    reproduces torch-CPU's grid_sample + norm bit for bit, cotr_dense_merge Pillow's mode-'F' resize) - against the golden
    output of the reference's own engine (cotr_flow maps by digest, then the task list drawn from them: correspondences and
    identifiers bit for bit), and against the same engine with Pillow crops + the host post-processing recipe."""
    from tests.engine_fixtures import CyclicFakeModel, digest
    from tests.test_zoom_engine_cpu import run_dense_case, FLOW_KEYS
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    img_a, img_b = synthetic_pair(int(g['meta'][0]))
    dev = ZoomEngine(CyclicFakeModel().cuda(), max_pairs=64)
    assert dev.make_dense_post.__name__ == '_DeviceDensePost' and dev.make_cropper.__name__ == '_DeviceCropper'
    host = ZoomEngine(CyclicFakeModel(), max_pairs=64, make_cropper=pil_cropper_factory,
                      make_dense_post=dense_post.host_dense_post_factory)
    flow_d, flow_h = dev.flow(img_a, img_b), host.flow(img_a, img_b)
    for k, d, h in zip(FLOW_KEYS, flow_d, flow_h):
        if 'resample' in k:                              # the warp runs through torch on the GPU in one, on the CPU in the other
            assert np.abs(d - h).max() <= 1e-2, k
            assert np.allclose(d[::9, ::9], g['flow_' + k], rtol=0, atol=1e-2), k
            continue
        assert np.array_equal(d, h), k
        assert np.array_equal(d[::9, ::9], g['flow_' + k]), k
        assert digest(d) == g['sha_' + k].tobytes(), k
    out_d, out_h = run_dense_case(g, dev), run_dense_case(g, host)
    for d, h in zip(out_d, out_h):
        assert np.array_equal(d, h)
    assert np.array_equal(out_d[0], g['corrs']) and np.array_equal(ids(out_d[1]), g['idx'])


def _dense_inputs(seed, rough):
    """pred [4,256,512,2] on the device for a non-square pair (2 x 2 overlapping patches) + geometry."""
    from tests.engine_fixtures import CyclicFakeModel
    img_a, img_b = synthetic_pair(seed)
    eng = ZoomEngine(CyclicFakeModel().cuda())
    pa, pb = eng._square_patches(img_a), eng._square_patches(img_b)
    pairs = [(i, j) for i in pa for j in pb]
    jj, ii = np.meshgrid(np.arange(512), np.arange(256))
    q = torch.from_numpy(np.stack([jj / 512, ii / 256], -1).reshape(1, -1, 2)).float().expand(4, -1, -1)
    rng = np.random.default_rng(seed)
    img = torch.from_numpy(rng.standard_normal((4, 3, 256, 512)).astype(np.float32))
    pred = CyclicFakeModel()(img, q)['pred_corrs'].view(4, 256, 512, 2).clone()
    if rough:           # answers far outside [0,1] (grid_sample's zero padding), on the border, exactly on grid nodes
        noise = torch.from_numpy(rng.standard_normal((4, 256, 512, 2)).astype(np.float32))
        pred[:, ::7, ::5] += noise[:, ::7, ::5]
        sub = pred[:, 3::11, 2::13]
        pred[:, 3::11, 2::13] = torch.from_numpy(rng.integers(-2, 514, tuple(sub.shape)).astype(np.float32)) / 512
        pred[0, 0, 0] = torch.tensor([5.0, -3.0])
    return img_a, img_b, pairs, pred.cuda()


@pytest.mark.parametrize('rough', [False, True])
def test_dense_cycle_kernel_vs_host_recipe(rough):
    """cotr_dense_cycle vs torch-CPU grid_sample + norm + numpy (oracle/dense_post.py <- inference_helper.py:137-158):
    bit for bit (the kernel follows the association and the FMA contractions of torch's CPU kernels)."""
    from cotr_amd.inference.zoom_engine import _DeviceDensePost, _patch_affines
    from cotr_amd import _lib
    img_a, img_b, pairs, pred = _dense_inputs(11, rough)
    lib = _lib.load_library()
    aff = np.stack([np.stack(_patch_affines(p_i, p_j, img_a.shape, img_b.shape)) for p_i, p_j in pairs])
    aff_d = torch.from_numpy(np.ascontiguousarray(aff)).cuda()
    maps = torch.empty((4, 256, 512, 3), device='cuda')
    _lib.check(lib.cotr_dense_cycle(pred.data_ptr(), 4, aff_d.data_ptr(), maps.data_ptr(), _lib.current_stream_ptr()),
               None, 'cotr_dense_cycle')
    got = maps.cpu().numpy()
    for k, (p_i, p_j) in enumerate(pairs):
        c_i, c_j = dense_post.cycle_maps(pred[k].cpu().numpy())
        t_i, t_j = dense_post.patch_affines(p_i, p_j, img_a.shape, img_b.shape)
        c_i[..., :2] = c_i[..., :2] @ t_i[:2, :2] + t_i[:, 2]
        c_j[..., :2] = c_j[..., :2] @ t_j[:2, :2] + t_j[:, 2]
        want = np.concatenate([c_i, c_j], axis=1)
        assert np.array_equal(got[k][..., :2], want[..., :2])             # re-centring + affine: exact
        assert np.array_equal(got[k][..., 2], want[..., 2], equal_nan=True)   # cycle error: torch-CPU's bits


def test_dense_cycle_kernel_special_coordinates():
    """NaN / inf / beyond-int-range / far-outside / exactly-on-the-border answers: the same bits (and the same NaNs) as
    torch-CPU's grid_sample + norm (a NaN or inf coordinate poisons the sample: the weights are multiplied into the zero the
    out-of-map neighbours contribute; a merely huge one samples zeros)."""
    from cotr_amd import _lib
    rng = np.random.default_rng(3)
    g = rng.random((2, 256, 512, 2), dtype=np.float32)
    special = [np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 0.0, 1.0, -1 / 512, 513 / 512, 0.5 / 512, 1 - 0.5 / 512,
               1 + 0.5 / 512, -0.5 / 512, -1.5 / 512, 2.0 ** 31 / 512, 5.0, -3.0]
    k = 0
    flat = g.reshape(-1, 2)
    for a in special:
        for b in [0.3, np.nan, np.inf, -1e30, 1 + 0.5 / 256, -0.5 / 256]:
            flat[k] = (a, b)
            flat[k + 7] = (b, a)
            k += 14
    lib = _lib.load_library()
    aff = torch.tensor([[[1.0, 0, 0, 0, 1, 0]] * 2] * 2, dtype=torch.float64).cuda()
    maps = torch.empty((2, 256, 512, 3), device='cuda')
    _lib.check(lib.cotr_dense_cycle(torch.from_numpy(g).cuda().data_ptr(), 2, aff.data_ptr(), maps.data_ptr(),
                                    _lib.current_stream_ptr()), None, 'cotr_dense_cycle')
    got = maps.cpu().numpy()[..., 2]
    for p in range(2):
        l, r = dense_post.cycle_maps(g[p])
        want = np.concatenate([l, r], axis=1)[..., 2]
        assert np.isnan(want).sum() > (50 if p == 0 else -1)          # the special values sit in the first pair
        assert np.array_equal(got[p], want, equal_nan=True)


@pytest.mark.parametrize('shapes', [((300, 420), (350, 330)), ((256, 256), (200, 390)), ((783, 1064), (1053, 689))])
def test_dense_merge_kernel_is_pillow_exact(shapes):
    """cotr_dense_merge on maps computed by the host recipe == float_image_resize (Pillow mode 'F', up- and
    down-scaling, identity) + merge_flow_patches of the reference, bit for bit."""
    from cotr_amd import _lib
    rng = np.random.default_rng(5)
    img_a = np.zeros(shapes[0] + (3,), np.uint8)
    img_b = np.zeros(shapes[1] + (3,), np.uint8)
    pa, pb = ZoomEngine._square_patches(img_a), ZoomEngine._square_patches(img_b)
    pairs = [(i, j) for i in pa for j in pb]
    n = len(pairs)
    maps = rng.standard_normal((n, 256, 512, 3)).astype(np.float32)
    maps[..., 2] = np.abs(maps[..., 2]) * 0.05
    maps[:, 100:140, :, 2] = 0.01                                          # ties between overlapping patches
    lib = _lib.load_library()
    maps_d = torch.from_numpy(maps).cuda()
    for side, shape in ((0, img_a.shape), (1, img_b.shape)):
        entries = []
        for k, pr in enumerate(pairs):
            x, y, s = pr[side]
            half = np.ascontiguousarray(maps[k][:, side * 256:(side + 1) * 256])
            entries.append((dense_post.float_image_resize(half, (s, s)), x, y, s, s, shape[1], shape[0]))
        want_flow, want_conf, _ = dense_post.merge_flow_patches(entries)
        boxes = torch.tensor([list(pr[side]) for pr in pairs], dtype=torch.int32).cuda()
        flow = torch.empty((shape[0], shape[1], 2), device='cuda')
        conf = torch.empty((shape[0], shape[1]), device='cuda')
        _lib.check(lib.cotr_dense_merge(maps_d.data_ptr(), boxes.data_ptr(), n, side, shape[0], shape[1], flow.data_ptr(),
                                        conf.data_ptr(), _lib.current_stream_ptr()), None, 'cotr_dense_merge')
        assert np.array_equal(conf.cpu().numpy().astype(np.float64), want_conf)
        assert np.array_equal(flow.cpu().numpy().astype(np.float64), want_flow)


def test_device_dense_post_end_to_end():
    """Default ZoomEngine (device crops + device post-processing) vs the host recipe on the same prediction: all four maps
    bit for bit, hence the same confident-pixel mask, the same random draw and the same correspondences."""
    from tests.engine_fixtures import CyclicFakeModel
    img_a, img_b = synthetic_pair(12)
    dev = ZoomEngine(CyclicFakeModel().cuda())
    host = ZoomEngine(CyclicFakeModel().cuda(), make_dense_post=dense_post.host_dense_post_factory)
    d, h = dev.flow(img_a, img_b), host.flow(img_a, img_b)
    for k in (0, 1, 3, 4):
        assert np.array_equal(d[k], h[k]), k
    np.random.seed(0)
    corrs = dev.cotr_corr_multiscale_with_cycle_consistency(img_a, img_b, ZOOMS, 1, max_corrs=20)
    np.random.seed(0)
    corrs_h = host.cotr_corr_multiscale_with_cycle_consistency(img_a, img_b, ZOOMS, 1, max_corrs=20)
    assert np.array_equal(corrs, corrs_h)
    assert corrs.shape == (20, 4) and np.isfinite(corrs).all()
    assert (corrs[:, 0] < img_a.shape[1]).all() and (corrs[:, 2] < img_b.shape[1]).all() and (corrs > 0).all()


def test_dense_pass_through_the_real_model():
    """The dense pass is ONE model call of [pairs, 131072, 2] queries (inference_helper.py:116-127, LARGE_GPU path);
    its rows must agree with the row-by-row calls of the reference's LARGE_GPU=False path (same crops, 512 queries)."""
    sd = synth_state_dict(0)
    img_a, img_b = synthetic_pair(6)
    hip = build_model(cotr_amd.default_args()).cuda().eval()
    hip.load_state_dict(sd)
    seen = {}

    class Spy(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, img, qs):
            out = self.m(img, qs)
            seen['img'], seen['q'], seen['out'] = img.clone(), qs.clone(), out['pred_corrs'].clone()
            return out

    eng = ZoomEngine(Spy(hip))
    corr_a, con_a, res_a, corr_b, con_b, res_b = eng.flow(img_a, img_b)
    assert corr_a.shape == img_a.shape[:2] + (2,) and con_b.shape == img_b.shape[:2]
    assert res_a.shape == img_a.shape and np.isfinite(corr_a).all() and np.isfinite(con_a).all()
    assert seen['q'].shape == (4, 131072, 2)
    big = seen['out'].view(4, 256, 512, 2)
    for row in (0, 97, 255):
        q_row = seen['q'].view(4, 256, 512, 2)[:, row].contiguous()
        small = hip(seen['img'], q_row)['pred_corrs']
        assert (big[:, row] - small).abs().max().item() * 256 < 1e-3      # px in the 256x512 network frame


@pytest.mark.parametrize('src_shape,dst_shape', [((420, 420), (300, 420)), ((350, 350), (350, 330)), ((256, 512), (700, 300)),
                                                 ((64, 48), (64, 48)), ((300, 200), (37, 411))])
@pytest.mark.parametrize('channels', [1, 2])
def test_resize_f32_is_pillow_exact(src_shape, dst_shape, channels):
    """cotr_resize_f32 == utils.float_image_resize (Pillow mode 'F' BILINEAR; utils.py:69-83), bit for bit: the map
    resize of SparseEngine's 'stretching' mode (sparse_engine.py:124-129)."""
    from cotr_amd.inference.zoom_engine import _DeviceDensePost
    rng = np.random.default_rng(src_shape[0] + dst_shape[1] + channels)
    arr = rng.standard_normal(src_shape + (channels,)).astype(np.float32)
    arr[::7, ::5] = 100.0
    want = dense_post.float_image_resize(arr, dst_shape)
    got = _DeviceDensePost(torch.device('cuda:0')).resize(arr, dst_shape)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    if channels == 1:
        assert np.array_equal(_DeviceDensePost(torch.device('cuda:0')).resize(arr[..., 0], dst_shape), want[..., 0])


def test_stretching_mode_on_device():
    """mode='stretching' with device crops + device maps vs the host recipe: maps to 2e-6, then the whole default path
    runs and returns in-bounds correspondences."""
    from tests.engine_fixtures import CyclicFakeModel
    img_a, img_b = synthetic_pair(13)
    dev = ZoomEngine(CyclicFakeModel().cuda(), mode='stretching')
    np.random.seed(0)
    corrs = dev.cotr_corr_multiscale(img_a, img_b, ZOOMS, 1, max_corrs=25)
    assert corrs.shape == (25, 4) and np.isfinite(corrs).all() and (corrs > 0).all()
    assert (corrs[:, 0] < img_a.shape[1]).all() and (corrs[:, 1] < img_a.shape[0]).all()
    assert (corrs[:, 2] < img_b.shape[1]).all() and (corrs[:, 3] < img_b.shape[0]).all()
    with pytest.raises(ValueError):
        ZoomEngine(CyclicFakeModel().cuda(), mode='pyramid')


# ---- FasterSparseEngine on the device (sparse_engine.py:267-427) ------------------------------------------------------------
@pytest.mark.parametrize('name', ['engine_faster_known', 'engine_faster_dense', 'engine_faster_dense_q', 'engine_faster_force_q'])
def test_faster_sparse_engine_with_device_crops_matches_the_reference_class(name, golden_dir):
    """The goldens of the reference's own FasterSparseEngine (pilots, squads, np.random.permutation order, zero-padded grouped
    model calls, the fallback loop) reproduced with every crop pair cut by the HIP kernel (Pillow-exact, so the fake model sees
    the reference's pixels) AND the dense initial pass post-processed on the device (bit-exact cycle-error maps, so the
    'confident' mask and the np.random.choice draw behind it are the reference's): correspondences, identifiers, crop
    bookkeeping and the model-call shapes, bit for bit."""
    from tests.test_zoom_engine_cpu import FASTER_CASES, run_faster_case
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    (corrs, idx), eng, model = run_faster_case(name, g, False, on_device=True)
    assert next(model.parameters()).is_cuda and eng.make_cropper.__name__ == '_DeviceCropper'
    assert eng.make_dense_post.__name__ == '_DeviceDensePost'
    assert np.array_equal(np.asarray(corrs, dtype=np.float64).reshape(-1, 4), g['corrs'])
    assert np.array_equal(ids(idx), g['idx'])
    assert eng.total_tasks - 4 == int(g['total_tasks'])
    known = FASTER_CASES[name][6]
    ref_calls = [(int(r[0]), int(r[5])) for r in g['calls']][8 if known else 4:]
    own_calls = [(a[0], b[1]) for a, b in model.calls][2 if known else 1:]
    assert own_calls == ref_calls
    assert any(q > 1 for _, q in own_calls)                  # squads really formed: grouped calls with several queries per pilot


def test_grouped_call_of_32_pilots_x_257_queries_through_the_real_model():
    """infer_batch_grouped at its largest shape (sparse_engine.py:277-282, 295-369): 32 pilots, each with max_load = 256 riders
    -> one model call img[32,3,256,512] x q[32,257,2] on crops cut by the device kernel.  Squads are forced by construction
    (32 well separated clusters of 257 tasks within a few pixels of each other); the HIP model's answers are checked against
    the CPU oracle on the same crops for three of the pilots, and every task steps with its own answer."""
    from cotr_amd.inference import FasterSparseEngine
    from cotr_amd.inference.zoom_engine import ZoomTask
    sd = synth_state_dict(0)
    hip = build_model(cotr_amd.default_args()).cuda().eval()
    hip.load_state_dict(sd)
    img_a, img_b = synthetic_pair(9, shape_a=(640, 640), shape_b=(640, 640))
    rng = np.random.default_rng(2)
    zoom = 0.125                                              # 80-pixel crops: the central half is +-20 px around the pilot
    centres = [(60 + 90 * (c % 6) + 40, 60 + 90 * (c // 6) + 30) for c in range(32)]
    tasks = []
    for cx, cy in centres:
        for _ in range(257):
            f = np.array([cx, cy]) + rng.uniform(-3, 3, 2)
            t = np.array([cx + 7, cy - 5]) + rng.uniform(-3, 3, 2)
            tasks.append(ZoomTask(img_a.shape, img_b.shape, f, t, 1.0, 1.0, 1, [zoom]))
    eng = FasterSparseEngine(hip, 32, mode='tile', max_load=256)
    np.random.seed(4)
    squads, boxes, queries = eng._form_grouped_batch(zoom, tasks)
    assert len(squads) == 32 and all(len(m) == 257 for m in squads) and queries.shape == (32, 257, 2)
    assert all(t.submitted for t in tasks)
    device = next(hip.parameters()).device
    cropper = eng.make_cropper(img_a, img_b, device)
    out = eng._forward(cropper, boxes, queries, device, count=False)
    assert out.shape == (32, 257, 2) and np.isfinite(out).all()
    buf = torch.empty((32, 3, 256, 512), dtype=torch.float32, device=device)
    crops = cropper(boxes, buf).cpu()
    for i in (0, 13, 31):
        ref = cotr_oracle.cotr_forward(sd, crops[i:i + 1], torch.from_numpy(queries[i:i + 1]))
        assert cotr_oracle.px_err(torch.from_numpy(out[i:i + 1]), ref) < 1e-3, i
    for i, members in enumerate(squads):                      # the loop body of cotr_corr_multiscale (:389-393)
        for j, t in enumerate(members):
            t.step(out[i, j])
    assert all(not t.submitted and t.total_iter == 1 for t in tasks)
