"""ZoomEngine's state machine against golden trajectories of the REFERENCE engine (SparseEngine + RefinementTask,
run unchanged in the authoring container by tests/golden/make_engine_golden.py) - same fake model, same synthetic
pair, crops by Pillow on the host.  Bit-exact: the engine logic is integer crop boxes + float32/float64 numpy."""
import os

import numpy as np
import pytest
import torch

from cotr_amd.inference import ZoomEngine, patch_boxes
from oracle.dense_post import host_dense_post_factory
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


# ---- default path: dense initial pass + task generation + early exit + cycle-consistency wrapper ---------------------
DENSE_CASES = ['engine_dense_default', 'engine_dense_default_c3', 'engine_dense_queries_filter',
               'engine_dense_queries_force', 'engine_cycle_default', 'engine_cycle_queries',
               'engine_stretch_default', 'engine_stretch_cycle']
FLOW_KEYS = ('corr_a', 'con_a', 'resample_a', 'corr_b', 'con_b', 'resample_b')


def run_dense_case(g, eng):
    seed, max_corrs, conv, nq, force, cycle = (int(v) for v in g['meta'][:6])
    img_a, img_b = synthetic_pair(seed)
    queries = None if nq < 0 else g['queries']
    np.random.seed(seed)                                 # gen_tasks draws from numpy's global RNG like the reference
    if cycle:
        return eng.cotr_corr_multiscale_with_cycle_consistency(img_a, img_b, ZOOMS, conv, max_corrs=max_corrs,
                                                               queries_a=queries, return_idx=True, return_cycle_error=True)
    return eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=max_corrs, queries_a=queries, return_idx=True,
                                    force=bool(force))


@pytest.mark.parametrize('name', DENSE_CASES[:2])
def test_flow_matches_reference_cotr_flow(name, golden_dir):
    """ZoomEngine.flow (engine plumbing) + oracle/dense_post.py (the host restatement of the post-processing) vs
    cotr_flow (inference_helper.py:104-182) of the reference: whole maps by digest."""
    from tests.engine_fixtures import CyclicFakeModel, digest
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    img_a, img_b = synthetic_pair(int(g['meta'][0]))
    model = CyclicFakeModel()
    out = ZoomEngine(model, make_cropper=pil_cropper_factory, make_dense_post=host_dense_post_factory).flow(img_a, img_b)
    for k, v in zip(FLOW_KEYS, out):
        assert np.array_equal(v[::9, ::9], g['flow_' + k]), k
        assert digest(v) == g['sha_' + k].tobytes(), k
    assert model.calls == [((4, 3, 256, 512), (4, 131072, 2))]     # all four patch pairs in ONE model call


@pytest.mark.parametrize('name', DENSE_CASES)
@pytest.mark.parametrize('max_pairs', [256, 40])
def test_default_path_matches_reference_engine(name, max_pairs, golden_dir):
    """cotr_corr_multiscale / ..._with_cycle_consistency without ``areas``: same correspondences, in the same order,
    as SparseEngine(model, 32, 'tile') - including where its group-of-32 loop stops (sparse_engine.py:208-218)."""
    from tests.engine_fixtures import CyclicFakeModel
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    mode = 'stretching' if len(g['meta']) > 6 and int(g['meta'][6]) else 'tile'
    eng = ZoomEngine(CyclicFakeModel(), max_pairs=max_pairs, make_cropper=pil_cropper_factory,
                     make_dense_post=host_dense_post_factory, mode=mode)
    out = run_dense_case(g, eng)
    assert np.array_equal(out[0], g['corrs'])
    assert np.array_equal(out[1], g['idx'])
    if len(out) == 3:
        assert np.array_equal(out[2], g['cycle_error'])


def test_reference_schedule_emulation():
    """_reference_schedule against a literal simulation of the reference loop."""
    from cotr_amd.inference.zoom_engine import _reference_schedule
    rng = np.random.default_rng(0)
    for trial in range(50):
        n = int(rng.integers(0, 200))
        steps = rng.integers(1, 5, n)
        good = rng.random(n) < 0.6
        max_corrs = int(rng.integers(1, 80))
        status = np.zeros(n, dtype=np.int64)             # literal version: per-task counters, list order
        while True:
            num_g = int((good & (status == steps)).sum())
            batch = [i for i in range(n) if status[i] < steps[i]][:32]
            if not batch or num_g >= max_corrs:
                break
            for i in batch:
                status[i] += 1
        want = status == steps
        assert np.array_equal(_reference_schedule(steps, good, 32, max_corrs, n), want)
        # with only a prefix known it either asks for more or gives the prefix of the full answer
        for known in range(0, n, 37):
            got = _reference_schedule(steps[:known], good[:known], 32, max_corrs, n)
            assert got is None or np.array_equal(got, want[:known])


def test_reference_constructor_signatures(golden_dir):
    """SparseEngine(model, 32, mode='tile') / FasterSparseEngine(model, 32, 'tile', max_load=256) as the demos build them
    (demo_single_pair.py:35, demo_reconstruction.py) give the reference SparseEngine's golden result."""
    from cotr_amd.inference import FasterSparseEngine, SparseEngine
    from tests.engine_fixtures import CyclicFakeModel
    g = np.load(os.path.join(golden_dir, 'engine_cycle_default.npz'))
    for eng in (SparseEngine(CyclicFakeModel(), 32, mode='tile'), FasterSparseEngine(CyclicFakeModel(), 32, 'tile', max_load=256)):
        eng.make_cropper, eng.make_dense_post = pil_cropper_factory, host_dense_post_factory
        out = run_dense_case(g, eng)
        assert np.array_equal(out[0], g['corrs']) and np.array_equal(out[2], g['cycle_error'])
    assert SparseEngine(CyclicFakeModel(), 32).mode == 'stretching'        # the reference's default (sparse_engine.py:18)


def test_default_cropper_fails_loudly_without_gpu_or_with_wrong_images():
    """No host fallback for the crop kernel: a CPU model raises; so do float / grey images (the reference's PIL path would
    reject them too)."""
    from cotr_amd import _lib
    eng = ZoomEngine(FakeModel())                       # parameters on the CPU -> device-side cropper refuses
    img_a, img_b = synthetic_pair(0)
    with pytest.raises(_lib.CotrHipError):
        eng.refine(img_a, img_b, [[10.0, 10.0]], [[12.0, 12.0]], 1.0, 1.0, [0.5], 1)
    with pytest.raises(ValueError):
        eng.refine(img_a.astype(np.float32), img_b, [[10.0, 10.0]], [[12.0, 12.0]], 1.0, 1.0, [0.5], 1)
    with pytest.raises(ValueError):
        eng.refine(img_a[..., 0], img_b, [[10.0, 10.0]], [[12.0, 12.0]], 1.0, 1.0, [0.5], 1)
