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
from tests.engine_fixtures import FakeModel, synthetic_pair  # noqa: E402

CASES = {                      # name: (seed, n_queries, converge_iters, force)
    'engine_c1_force': (0, 24, 1, True),
    'engine_c3_force': (1, 24, 3, True),
    'engine_c3_filter': (2, 40, 3, False),
}


def main():
    torch.set_num_threads(1)
    ref_import.import_reference_models()                 # installs the stubs, puts the reference on sys.path
    cwd = os.getcwd()
    os.chdir(ref_import.REFERENCE_ROOT)                  # COTR/global_configs asserts ./out and ./tb_out exist
    try:
        from COTR.inference.sparse_engine import SparseEngine
        from COTR.inference.inference_helper import cotr_corr_base
    finally:
        os.chdir(cwd)
    zoom_ins = np.linspace(0.5, 0.0625, 4)               # demo_single_pair.py:37
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
