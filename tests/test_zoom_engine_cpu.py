"""ZoomEngine's state machine against golden trajectories of the REFERENCE engine (SparseEngine + RefinementTask,
run unchanged in the authoring container by tests/golden/make_engine_golden.py) - same fake model, same synthetic
pair, crops by Pillow on the host.  Bit-exact: the engine logic is integer crop boxes + float32/float64 numpy."""
import os

import numpy as np
import pytest
import torch

from cotr_amd.inference import ZoomEngine, patch_boxes
from oracle.dense_post import host_dense_post_factory
from tests.engine_fixtures import FakeModel, ids, synthetic_pair, pil_cropper_factory

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
    if force:      # known areas demand force=True (sparse_engine.py:110); the filtered conclusion of the same tasks is
        corrs = eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=n, queries_a=g['init'][:, :2],   # checked in
                                         force=True, areas=[1.0, 1.0], init_b=g['init'][:, 2:])   # test_return_tasks_only_*
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
    assert np.array_equal(ids(out[1]), g['idx'])
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
    """SparseEngine(model, 32, mode='tile') as the demos build it (demo_single_pair.py:35) gives the reference
    SparseEngine's golden result; FasterSparseEngine has its own goldens below."""
    from cotr_amd.inference import FasterSparseEngine, SparseEngine
    from tests.engine_fixtures import CyclicFakeModel
    g = np.load(os.path.join(golden_dir, 'engine_cycle_default.npz'))
    eng = SparseEngine(CyclicFakeModel(), 32, mode='tile')
    eng.make_cropper, eng.make_dense_post = pil_cropper_factory, host_dense_post_factory
    out = run_dense_case(g, eng)
    assert np.array_equal(out[0], g['corrs']) and np.array_equal(out[2], g['cycle_error'])
    assert SparseEngine(CyclicFakeModel(), 32).mode == 'stretching'        # the reference's default (sparse_engine.py:18)
    assert FasterSparseEngine(CyclicFakeModel(), 32, 'tile', max_load=256).max_load == 256       # demo_reconstruction.py


FASTER_CASES = {   # tests/golden/make_engine_golden.py FASTER_CASES: (seed, model, nq, max_corrs, conv, force, known, batch, load)
    'engine_faster_known': (11, 'fake', 150, 150, 2, True, True, 8, 6),
    'engine_faster_dense': (12, 'cyclic', None, 60, 1, False, False, 32, 256),
    'engine_faster_dense_q': (13, 'cyclic', 90, 70, 3, False, False, 16, 4),
    'engine_faster_force_q': (14, 'cyclic', 64, 64, 1, True, False, 4, 256),
}


def run_faster_case(name, g, tasks_only, **engine_kw):
    from cotr_amd.inference import FasterSparseEngine
    from tests.engine_fixtures import CyclicFakeModel
    seed, kind, nq, max_corrs, conv, force, known, bs, load = FASTER_CASES[name]
    img_a, img_b = synthetic_pair(seed)
    model = FakeModel() if kind == 'fake' else CyclicFakeModel()
    if engine_kw.pop('on_device', False):        # parameters on the GPU: the engine cuts its crops with the HIP kernel
        model = model.cuda()
    eng = FasterSparseEngine(model, bs, mode='tile', max_load=load)
    for k, v in engine_kw.items():
        setattr(eng, k, v)
    np.random.seed(seed)
    res = eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=max_corrs, queries_a=None if nq is None else g['queries'],
                                   return_idx=True, force=force, areas=[1.0, 1.0] if known else None,
                                   return_tasks_only=tasks_only)
    return res, eng, model


@pytest.mark.parametrize('name', list(FASTER_CASES))
def test_faster_sparse_engine_matches_the_reference_class(name, golden_dir):
    """FasterSparseEngine (sparse_engine.py:267-427: pilots, squads of nearby tasks answered from the pilot's crops,
    np.random.permutation pilot order, zero-padded query batches, early give-up per level, fallback loop on the last
    level) against goldens of the reference's own class with the same fake model: same correspondences, identifiers,
    crop bookkeeping, the same sequence of model-call shapes, and the same state of EVERY task at the end (including the
    tasks the reference's loop leaves unfinished at an earlier level)."""
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    kw = dict(make_cropper=pil_cropper_factory, make_dense_post=host_dense_post_factory)
    (corrs, idx), eng, model = run_faster_case(name, g, False, **kw)
    assert np.array_equal(np.asarray(corrs, dtype=np.float64).reshape(-1, 4), g['corrs'])
    assert np.array_equal(ids(idx), g['idx'])
    assert eng.total_tasks - (4 if FASTER_CASES[name][6] else 4) == int(g['total_tasks'])   # + our 4 patch pairs of the init pass
    # model calls of the zoom loops (after the initial pass, which is batched differently here): [pairs, queries per pair]
    ref_calls = [(int(r[0]), int(r[5])) for r in g['calls']][8 if FASTER_CASES[name][6] else 4:]
    own_calls = [(a[0], b[1]) for a, b in model.calls][2 if FASTER_CASES[name][6] else 1:]
    assert own_calls == ref_calls
    tasks, _, _ = run_faster_case(name, g, True, **kw)
    assert np.array_equal(np.array([t.status == 'finished' for t in tasks]), g['task_status'])
    assert np.array_equal(np.array([t.submitted for t in tasks]), g['task_submitted'])
    assert np.array_equal(np.array([t.total_iter for t in tasks]), g['task_iters'])
    assert np.array_equal(np.array([len(t.loc_history) for t in tasks]), g['task_levels'])
    assert np.array_equal(np.array([t.best_loc_to for t in tasks], dtype=np.float64), g['task_best'])
    assert np.array_equal(np.array([t.loc_from for t in tasks], dtype=np.float64), g['task_from'])


@pytest.mark.parametrize('name', CASES)
def test_return_tasks_only_gives_the_reference_task_states(name, golden_dir):
    """cotr_corr_multiscale(..., return_tasks_only=True) (sparse_engine.py:218-219): ZoomTask objects carrying the
    reference RefinementTask's loc_history / best_loc_to / status, concluded like SparseEngine.conclude_tasks."""
    from cotr_amd.inference.zoom_engine import conclude_tasks
    torch.set_num_threads(1)
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    seed, n, conv, force = (int(v) for v in g['meta'])
    img_a, img_b = synthetic_pair(seed)
    eng = ZoomEngine(FakeModel(), max_pairs=16, make_cropper=pil_cropper_factory)
    tasks = eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=n, queries_a=g['init'][:, :2], force=True,
                                     areas=[1.0, 1.0], init_b=g['init'][:, 2:], return_tasks_only=True)
    assert len(tasks) == n and all(t.status == 'finished' and t.identifier is None for t in tasks)
    assert np.array_equal(np.array([np.array(t.loc_history) for t in tasks]), g['loc_history'])
    assert np.array_equal(np.array([t.best_loc_to for t in tasks]), g['best'])
    corrs, _ = conclude_tasks(tasks, return_idx=True, force=bool(force), img_a_shape=img_a.shape[:2], img_b_shape=img_b.shape[:2])
    assert np.array_equal(corrs, g['corrs'])
    with pytest.raises(AssertionError):                      # the reference insists on force=True with known areas (:110)
        eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=n, queries_a=g['init'][:, :2], force=False,
                                 areas=[1.0, 1.0], init_b=g['init'][:, 2:])


def test_return_tasks_only_after_an_early_exit(golden_dir):
    """Default path stopping early (max_corrs good tasks reached): the task list holds finished, partially stepped and
    untouched tasks exactly where the reference's group-of-32 loop leaves them; concluding it gives the golden rows."""
    from cotr_amd.inference.zoom_engine import conclude_tasks
    from tests.engine_fixtures import CyclicFakeModel
    torch.set_num_threads(1)
    for name in ('engine_dense_default', 'engine_dense_default_c3'):
        g = np.load(os.path.join(golden_dir, name + '.npz'))
        seed, max_corrs, conv = (int(v) for v in g['meta'][:3])
        img_a, img_b = synthetic_pair(seed)
        eng = ZoomEngine(CyclicFakeModel(), max_pairs=64, make_cropper=pil_cropper_factory, make_dense_post=host_dense_post_factory)
        np.random.seed(seed)
        tasks = eng.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=max_corrs, return_tasks_only=True)
        corrs, idx = conclude_tasks(tasks, True, False, img_a.shape[:2], img_b.shape[:2])
        assert np.array_equal(corrs[:max_corrs], g['corrs'])
        states = {t.status for t in tasks}
        assert 'finished' in states and 'unfinished' in states
        for t in tasks:                                      # internal consistency of the partially stepped ones
            assert len(t.loc_history) == (5 if t.status == 'finished' else min(t.total_iter, 3) + 1)


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


@pytest.mark.parametrize('kind', ['unit', 'outside', 'near_identity', 'special'])
def test_c_cycle_restatement_equals_torch_cpu(kind):
    """oracle/dense_cycle_ref.c (the plain-C statement of torch-CPU's grid_sample + norm association, which
    cotr_amd/csrc/dense_post.hip reproduces on the device) against the reference's own three lines run through torch CPU
    (oracle/dense_post.cycle_maps <- inference_helper.py:137-139): the cycle-error map bit for bit, NaNs in the same places."""
    from oracle.build_c import dense_cycle_ref
    from oracle.dense_post import cycle_maps
    rng = np.random.default_rng(17)
    if kind == 'unit':
        g = rng.random((256, 512, 2), dtype=np.float32)
    elif kind == 'outside':                                   # far outside [0,1]: zero padding, the -1 border cell
        g = (rng.random((256, 512, 2), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    elif kind == 'near_identity':                             # what a trained model answers: the query grid + noise
        jj, ii = np.meshgrid(np.arange(512), np.arange(256))
        g = (np.stack([jj / 512, ii / 256], -1) + rng.normal(0, 0.01, (256, 512, 2))).astype(np.float32)
    else:
        g = rng.random((256, 512, 2), dtype=np.float32)
        flat = g.reshape(-1, 2)
        special = [np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 0.0, 1.0, -1 / 512, 513 / 512, 0.5 / 512, 1 - 0.5 / 512,
                   1 + 0.5 / 512, -0.5 / 512, -1.5 / 512, 2.0 ** 31 / 512, 5.0, -3.0]
        k = 0
        for a in special:
            for b in [0.3, np.nan, np.inf, -1e30, 1 + 0.5 / 256, -0.5 / 256]:
                flat[k] = (a, b)
                flat[k + 7] = (b, a)
                k += 14
    _, err = dense_cycle_ref(g[None])
    left, right = cycle_maps(g)
    want = np.concatenate([left, right], axis=1)[..., 2]
    assert np.array_equal(err[0], want, equal_nan=True)
    if kind == 'special':
        assert np.isnan(want).sum() > 50
