"""Golden trajectories of the REFERENCE's own zoom-in engine (COTR/inference/sparse_engine.py + refinement_task.py,
imported unchanged behind the stubs of oracle/ref_import.py) driven by the deterministic fake model of
tests/engine_fixtures.py on a synthetic pair.  Authoring container only (needs /root/reference).

    python tests/golden/make_engine_golden.py      # rewrites tests/golden/engine_*.npz
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from tests.engine_fixtures import CyclicFakeModel, FakeModel, digest, synthetic_pair  # noqa: E402

CASES = {                      # name: (seed, n_queries, converge_iters, force)
    'engine_c1_force': (0, 24, 1, True),
    'engine_c3_force': (1, 24, 3, True),
    'engine_c3_filter': (2, 40, 3, False),
}


# default path (dense initial pass, sparse_engine.py:116-195): name: (seed, max_corrs, converge_iters, queries, force, cycle)
DENSE_CASES = {
    'engine_dense_default': (3, 40, 1, None, False, False),        # 80 tasks, stops early once 40 are good
    'engine_dense_default_c3': (4, 25, 3, None, False, False),     # tasks need 4..6 steps: groups of 32 interleave
    'engine_dense_queries_filter': (5, 30, 2, 50, False, False),
    'engine_dense_queries_force': (6, 30, 1, 30, True, False),
    'engine_cycle_default': (7, 15, 1, None, False, True),
    'engine_cycle_queries': (8, 12, 2, 60, False, True),
    'engine_stretch_default': (9, 30, 1, None, False, False, 'stretching'),     # sparse_engine.py:114-129
    'engine_stretch_cycle': (10, 12, 1, None, False, True, 'stretching'),
}


def dense_goldens(SparseEngine, cotr_flow, zoom_ins):
    for name, case in DENSE_CASES.items():
        seed, max_corrs, conv, nq, force, cycle = case[:6]
        mode = case[6] if len(case) > 6 else 'tile'
        img_a, img_b = synthetic_pair(seed)
        rng = np.random.default_rng(seed + 100)
        queries = None
        if nq is not None:    # some outside the image on purpose (skipped by the reference unless force)
            queries = np.stack([rng.uniform(-8, img_a.shape[1] + 8, nq), rng.uniform(-8, img_a.shape[0] + 8, nq)], 1)
        model = CyclicFakeModel()
        out = {}
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            flow = cotr_flow(model, img_a, img_b)
            engine = SparseEngine(model, 32, mode=mode)
            np.random.seed(seed)
            if cycle:
                corrs, idx, err = engine.cotr_corr_multiscale_with_cycle_consistency(
                    img_a, img_b, zoom_ins, conv, max_corrs=max_corrs, queries_a=queries, return_idx=True,
                    return_cycle_error=True)
                out['cycle_error'] = err
            else:
                corrs, idx = engine.cotr_corr_multiscale(img_a, img_b, zoom_ins, conv, max_corrs=max_corrs,
                                                         queries_a=queries, return_idx=True, force=force)
        idx = np.array([-1 if i is None else i for i in idx], dtype=np.int64)
        for k, v in zip(('corr_a', 'con_a', 'resample_a', 'corr_b', 'con_b', 'resample_b'), flow):
            out['flow_' + k] = np.ascontiguousarray(v[::9, ::9])             # a sample for diagnostics ...
            out['sha_' + k] = np.frombuffer(digest(v), dtype=np.uint8)       # ... and a digest of the whole map
        np.savez_compressed(os.path.join(HERE, name + '.npz'), corrs=corrs, idx=idx,
                            queries=np.zeros((0, 2)) if queries is None else queries,
                            meta=np.array([seed, max_corrs, conv, -1 if nq is None else nq, int(force), int(cycle), int(mode == 'stretching')]),
                            total_tasks=np.array(engine.total_tasks), **out)
        print(name, 'kept', len(corrs), 'crops', engine.total_tasks, 'confident a/b',
              float((flow[1] < 0.02).mean()), float((flow[4] < 0.02).mean()))


# FasterSparseEngine (sparse_engine.py:267-427): name: (seed, model, n_queries | None, max_corrs, converge_iters, force,
#                                                       known scale, batch_size, max_load)
FASTER_CASES = {
    'engine_faster_known': (11, 'fake', 150, 150, 2, True, True, 8, 6),          # squads of <= 7 on 8 pilots per call
    'engine_faster_dense': (12, 'cyclic', None, 60, 1, False, False, 32, 256),   # demo settings (max_load 256)
    'engine_faster_dense_q': (13, 'cyclic', 90, 70, 3, False, False, 16, 4),     # filtered queries, converge_iters 3
    'engine_faster_force_q': (14, 'cyclic', 64, 64, 1, True, False, 4, 256),     # tiny batch: grouped loop gives up early
}


def faster_case_inputs(name):
    seed, kind, nq = FASTER_CASES[name][:3]
    img_a, img_b = synthetic_pair(seed)
    rng = np.random.default_rng(seed + 100)
    queries = None
    if nq is not None:
        if FASTER_CASES[name][6]:      # known scale: clustered queries inside the image (several tasks per crop centre)
            centres = np.stack([rng.uniform(40, img_a.shape[1] - 40, 12), rng.uniform(40, img_a.shape[0] - 40, 12)], 1)
            queries = centres[rng.integers(0, 12, nq)] + rng.normal(0, 6.0, (nq, 2))
            queries = np.clip(queries, 5, [img_a.shape[1] - 5, img_a.shape[0] - 5])
        else:
            queries = np.stack([rng.uniform(-8, img_a.shape[1] + 8, nq), rng.uniform(-8, img_a.shape[0] + 8, nq)], 1)
    return img_a, img_b, queries


def faster_goldens(FasterSparseEngine, zoom_ins):
    for name, (seed, kind, nq, max_corrs, conv, force, known, bs, load) in FASTER_CASES.items():
        img_a, img_b, queries = faster_case_inputs(name)
        out = {}
        for tasks_only in (False, True):
            model = FakeModel() if kind == 'fake' else CyclicFakeModel()
            engine = FasterSparseEngine(model, bs, mode='tile', max_load=load)
            np.random.seed(seed)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                res = engine.cotr_corr_multiscale(img_a, img_b, zoom_ins, conv, max_corrs=max_corrs,
                                                  queries_a=None if queries is None else queries.copy(), return_idx=True,
                                                  force=force, areas=[1.0, 1.0] if known else None,
                                                  return_tasks_only=tasks_only)
            if tasks_only:
                out['task_status'] = np.array([t.status == 'finished' for t in res])
                out['task_submitted'] = np.array([bool(t.submitted) for t in res])
                out['task_iters'] = np.array([t.total_iter for t in res])
                out['task_levels'] = np.array([len(t.loc_history) for t in res])
                out['task_best'] = np.array([np.asarray(t.best_loc_to, dtype=np.float64) for t in res])
                out['task_from'] = np.array([np.asarray(t.loc_from, dtype=np.float64) for t in res])
            else:
                corrs, idx = res
                out['corrs'] = np.asarray(corrs, dtype=np.float64).reshape(-1, 4)
                out['idx'] = np.array([-1 if i is None else i for i in idx], dtype=np.int64)
                out['total_tasks'] = np.array(engine.total_tasks)
                out['calls'] = np.array([list(a) + list(b) for a, b in model.calls], dtype=np.int64)   # [B,3,256,512,B,Q,2]
        np.savez_compressed(os.path.join(HERE, name + '.npz'),
                            queries=np.zeros((0, 2)) if queries is None else queries, **out)
        q_sizes = out['calls'][:, 5]
        print(name, 'tasks', len(out['task_status']), 'finished', int(out['task_status'].sum()), 'kept', len(out['corrs']),
              'model calls', len(out['calls']), 'grouped (Q>1)', int((q_sizes > 1).sum() - (q_sizes > 1000).sum()),
              'largest squad', int(q_sizes[q_sizes < 1000].max()), 'crops counted', int(out['total_tasks']))


def main():
    torch.set_num_threads(1)
    ref_import.import_reference_models()                 # installs the stubs, puts the reference on sys.path
    cwd = os.getcwd()
    os.chdir(ref_import.REFERENCE_ROOT)                  # COTR/global_configs asserts ./out and ./tb_out exist
    try:
        from COTR.inference.sparse_engine import FasterSparseEngine, SparseEngine
        from COTR.inference.inference_helper import cotr_corr_base, cotr_flow
    finally:
        os.chdir(cwd)
    zoom_ins = np.linspace(0.5, 0.0625, 4)               # demo_single_pair.py:37
    if '--faster-only' in sys.argv or ('--sparse-only' not in sys.argv and '--dense-only' not in sys.argv):
        faster_goldens(FasterSparseEngine, zoom_ins)
        if '--faster-only' in sys.argv:
            return
    if '--sparse-only' not in sys.argv:
        dense_goldens(SparseEngine, cotr_flow, zoom_ins)
    if '--dense-only' in sys.argv:
        return
    for name, (seed, n, conv, force) in CASES.items():
        img_a, img_b = synthetic_pair(seed)
        rng = np.random.default_rng(seed + 100)
        queries = np.stack([rng.uniform(5, img_a.shape[1] - 5, n), rng.uniform(5, img_a.shape[0] - 5, n)], 1)
        model = FakeModel()
        with contextlib.redirect_stdout(io.StringIO()):
            init = cotr_corr_base(model, img_a, img_b, queries.copy())        # [n,4]: what gen_tasks_w_known_scale uses
            engine = SparseEngine(model, 32, mode='tile')
            tasks = engine.cotr_corr_multiscale(img_a, img_b, zoom_ins, conv, max_corrs=n, queries_a=queries.copy(),
                                                force=True, areas=[1.0, 1.0], return_tasks_only=True)
            corrs, idx = engine.conclude_tasks(tasks, return_idx=True, force=force,
                                               img_a_shape=img_a.shape[:2], img_b_shape=img_b.shape[:2])
        hist = np.array([np.array(t.loc_history) for t in tasks])               # [n, levels+1, 2]
        best = np.array([t.best_loc_to for t in tasks])
        np.savez_compressed(os.path.join(HERE, name + '.npz'), queries=queries, init=init, loc_history=hist, best=best,
                            corrs=corrs, total_tasks=np.array(engine.total_tasks),
                            meta=np.array([seed, n, conv, int(force)]))
        iters = [t.total_iter for t in tasks]
        print(name, 'tasks', len(tasks), 'kept', len(corrs), 'iterations/task min..max', min(iters), max(iters),
              'crops', engine.total_tasks)


if __name__ == '__main__':
    main()
