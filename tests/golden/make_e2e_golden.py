"""End-to-end goldens: the REFERENCE's own engines (COTR/inference/sparse_engine.py SparseEngine / FasterSparseEngine, imported
unchanged behind the stubs of oracle/ref_import.py) driving the REFERENCE's own torch model (COTR.models.build_model, seeded random
weights of cotr_amd.utils.synth - there is no checkpoint offline) on CPU, exactly as demo_single_pair.py:25-45 drives them:
zoom levels np.linspace(0.5, 0.0625, 4), the recursive crop -> network -> re-centre loop, scale known (areas = [1, 1]) and force = True
(with random weights the dense initial pass of the default / cycle-consistency path finds no confident region and keeps no task:
`assert corr_f.shape[0] > 0`, sparse_engine.py:247 - those paths are pinned by the engine goldens on the fake model instead).
The stored data are inputs (seeds, queries) and the engines' outputs (correspondences, kept identifiers, crop counts); the GPU tests
run cotr_amd.inference's engines on the cotr_amd binding from the same inputs and must land on the same correspondences
(tests/test_e2e_reference_engines_gpu.py).  Authoring container only (needs /root/reference); takes a few minutes of CPU.

    python tests/golden/make_e2e_golden.py      # rewrites tests/golden/e2e_*.npz
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from cotr_amd.utils.synth import synth_state_dict  # noqa: E402
from tests.engine_fixtures import synthetic_pair  # noqa: E402

ZOOMS = np.linspace(0.5, 0.0625, 4)
# name: (engine, image seed, queries, converge_iters, cycle, batch_size, max_load)
CASES = {
    'e2e_sparse_known': ('sparse', 21, 12, 1, False, 32, None),       # SparseEngine.cotr_corr_multiscale, scale known (areas = [1, 1])
    'e2e_sparse_c2': ('sparse', 23, 8, 2, False, 4, None),            # converge_iters 2, batches of 4 (a ragged last batch)
    'e2e_faster_known': ('faster', 24, 40, 1, False, 8, 6),           # FasterSparseEngine: clustered queries -> squads on shared crops
}


def case_inputs(name):
    kind, seed, nq = CASES[name][:3]
    img_a, img_b = synthetic_pair(seed)
    rng = np.random.default_rng(seed + 500)
    if kind == 'faster':   # clustered: several queries share a pilot's crop (sparse_engine.py:339-369)
        centres = np.stack([rng.uniform(50, img_a.shape[1] - 50, 6), rng.uniform(50, img_a.shape[0] - 50, 6)], 1)
        q = centres[rng.integers(0, 6, nq)] + rng.normal(0, 5.0, (nq, 2))
        q = np.clip(q, 8, [img_a.shape[1] - 8, img_a.shape[0] - 8])
    else:
        q = np.stack([rng.uniform(8, img_a.shape[1] - 8, nq), rng.uniform(8, img_a.shape[0] - 8, nq)], 1)
    return img_a, img_b, q


def main():
    ref_import.import_reference_models()
    from COTR.inference.sparse_engine import SparseEngine, FasterSparseEngine
    torch.set_grad_enabled(False)
    model = ref_import.build_reference_model()
    model.load_state_dict(synth_state_dict(0))
    model.eval()
    for name, (kind, seed, nq, conv, cycle, bs, load) in CASES.items():
        img_a, img_b, queries = case_inputs(name)
        engine = SparseEngine(model, bs, mode='tile') if kind == 'sparse' else FasterSparseEngine(model, bs, mode='tile', max_load=load)
        np.random.seed(seed)
        t0 = time.time()
        out = {}
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            if cycle:
                corrs, idx, err = engine.cotr_corr_multiscale_with_cycle_consistency(
                    img_a, img_b, ZOOMS, conv, max_corrs=nq, queries_a=queries.copy(), return_idx=True, return_cycle_error=True)
                out['cycle_error'] = np.asarray(err)
            else:
                corrs, idx = engine.cotr_corr_multiscale(img_a, img_b, ZOOMS, conv, max_corrs=nq, queries_a=queries.copy(),
                                                         return_idx=True, force=True, areas=[1.0, 1.0])
        idx = np.array([-1 if i is None else i for i in idx], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), corrs=np.asarray(corrs), idx=idx, queries=queries,
                            meta=np.array([seed, nq, conv, int(cycle), bs, -1 if load is None else load]),
                            total_tasks=np.array(getattr(engine, 'total_tasks', -1)), **out)
        print(f'{name}: kept {len(corrs)} of {nq}, crops {getattr(engine, "total_tasks", -1)}, {time.time() - t0:.1f} s', flush=True)


if __name__ == '__main__':
    main()
