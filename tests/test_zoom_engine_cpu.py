"""ZoomEngine's state machine against golden trajectories of the REFERENCE engine (SparseEngine + RefinementTask,
run unchanged in the authoring container by tests/golden/make_engine_golden.py) - same fake model, same synthetic
pair, crops by Pillow on the host.  Bit-exact: the engine logic is integer crop boxes + float32/float64 numpy."""
import os

import numpy as np
import pytest
import torch

from cotr_amd.inference import ZoomEngine, patch_boxes
from tests.engine_fixtures import FakeModel, synthetic_pair, pil_cropper_factory

CASES = ['engine_c1_force', 'engine_c3_force', 'engine_c3_filter']
ZOOMS = np.linspace(0.5, 0.0625, 4)


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('max_pairs', [256, 7])
def test_trajectories_match_reference_engine(name, max_pairs, golden_dir):
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    seed, n, conv, force = (int(v) for v in g['meta'])
    img_a, img_b = synthetic_pair(seed)
    model = FakeModel()
    eng = ZoomEngine(model, max_pairs=max_pairs, make_cropper=pil_cropper_factory)
    res = eng.refine(img_a, img_b, g['init'][:, :2], g['init'][:, 2:], 1.0, 1.0, ZOOMS, conv, force=bool(force))
    assert np.array_equal(res.loc_history.transpose(1, 0, 2), g['loc_history'])      # every level, every task
    assert np.array_equal(res.loc_to, g['best'])
    assert res.crops == int(g['total_tasks'])                                         # same number of network inputs
    corrs = eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=n, queries_a=g['init'][:, :2],
                                     force=bool(force), areas=[1.0, 1.0], init_b=g['init'][:, 2:])
    assert np.array_equal(corrs, g['corrs'])
    # one model call per chunk per iteration instead of one per 32 tasks per task-iteration
    assert all(shape_q[1] == 1 for _, shape_q in model.calls)


def test_patch_boxes_is_get_patch_centered_at():
    """inference_helper.py:78-102 on scalars vs the vectorised version, including clamping at the borders."""
    shape = (301, 457, 3)
    rng = np.random.default_rng(0)
    pos = np.concatenate([rng.uniform(-30, 500, (200, 2)), [[0, 0], [456.9, 300.9], [228.5, 150.5]]])
    for scale in (1.0, 0.5, 0.37, 0.0625, 1.7):
        x, y, size = patch_boxes(shape, pos, scale)
        for i, p in enumerate(pos):
            h, w, _ = shape
            s = min(h, w) * float(np.clip(scale, 0.0, 1.0))
            s = int((s // 2) * 2)
            ly, lx = int(p[1] - s // 2), int(p[0] - s // 2)
            ly, lx = max(ly, 0), max(lx, 0)
            if ly + s > h:
                ly -= (ly + s) - h
            if lx + s > w:
                lx -= (lx + s) - w
            assert (x[i], y[i], size) == (lx, ly, s)


@pytest.mark.parametrize('name', CASES)
def test_corr_base_matches_reference(name, golden_dir):
    """cotr_corr_base (inference_helper.py:185-232) incl. the two-patch tiling of non-square images and the cycle
    selection; golden 'init' is the reference's own output with the same fake model."""
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    img_a, img_b = synthetic_pair(int(g['meta'][0]))
    eng = ZoomEngine(FakeModel(), make_cropper=pil_cropper_factory)
    assert np.array_equal(eng.corr_base(img_a, img_b, g['queries']), g['init'])
