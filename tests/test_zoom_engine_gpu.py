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
from oracle import cotr_oracle
from tests.engine_fixtures import FakeModel, synthetic_pair, pil_cropper_factory

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
def test_default_path_device_crops_equal_host_crops(name, golden_dir):
    """Whole default path with the HIP crop kernel == the same engine with Pillow crops on the host, bit for bit, and
    == the reference engine's golden output (the host part contains torch's CPU grid_sample, so the golden comparison
    allows for a different CPU's rounding: the maps to 1e-5, the final list exactly only if the maps were exact)."""
    from tests.engine_fixtures import CyclicFakeModel, digest
    from tests.test_zoom_engine_cpu import run_dense_case, FLOW_KEYS
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    img_a, img_b = synthetic_pair(int(g['meta'][0]))
    dev = ZoomEngine(CyclicFakeModel().cuda(), max_pairs=64)
    host = ZoomEngine(CyclicFakeModel(), max_pairs=64, make_cropper=pil_cropper_factory)
    flow_d, flow_h = dev.flow(img_a, img_b), host.flow(img_a, img_b)
    exact = True
    for k, d, h in zip(FLOW_KEYS, flow_d, flow_h):
        assert np.array_equal(d, h), k
        assert np.allclose(d[::9, ::9], g['flow_' + k], rtol=0, atol=1e-5 if 'resample' not in k else 1e-2), k
        exact &= digest(d) == g['sha_' + k].tobytes()
    out_d, out_h = run_dense_case(g, dev), run_dense_case(g, host)
    for d, h in zip(out_d, out_h):
        assert np.array_equal(d, h)
    if exact:
        assert np.array_equal(out_d[0], g['corrs']) and np.array_equal(out_d[1], g['idx'])


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
