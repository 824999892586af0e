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
