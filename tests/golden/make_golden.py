"""Generate the committed golden vectors from the REFERENCE ITSELF.

Runs only in the authoring container (needs ``/root/reference``): imports the
reference's ``COTR.models`` unchanged behind stubs (``oracle/ref_import.py``),
loads seeded synthetic weights (``cotr_amd/utils/synth.py`` - no trained
checkpoint exists offline), runs ``model(img, queries)`` in
fp32 and fp64 on CPU and stores the outputs (plus the encoder memory the
reference's ``Transformer.forward`` returns, sub-sampled) in
``tests/golden/<case>.npz``.

    python tests/golden/make_golden.py [case ...]  # rewrites every (or the named) case

The inputs/weights are NOT stored (74 MB): they are regenerated from the seeds
in the file by the numpy PCG64 generator; ``weights_checksum`` guards against
generator drift.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cotr_amd.utils.synth import synth_state_dict, synth_inputs, state_checksum  # noqa: E402
from oracle import ref_import  # noqa: E402

# name -> (weight seed, attn_gain, B, Q, input seed, query range)
CASES = {
    'primary_b1_q1000': (0, 1.0, 1, 1000, 1, (0.0, 1.0)),   # BASELINE.json configs[1] operating point
    'ragged_b2_q257': (0, 1.0, 2, 257, 2, (0.0, 1.0)),      # FasterSparseEngine max_load+1 (sparse_engine.py:363-366)
    'single_b1_q1': (0, 1.0, 1, 1, 3, (0.0, 1.0)),
    'engine_b4_q1': (0, 1.0, 4, 1, 4, (0.0, 1.0)),          # SparseEngine.infer_batch shape (sparse_engine.py:47-56)
    'peaky_b1_q64': (7, 4.0, 1, 64, 5, (0.0, 1.0)),         # sharp softmax
    'outside_b1_q96': (0, 1.0, 1, 96, 6, (-0.5, 1.5)),      # cycle pass feeds unbounded predictions back as queries
    # softmax extremes for the exp2-domain attention kernel: logits x16^2 / x32^2 (near one-hot rows, large negative
    # exponents after the running-max subtraction) and q = k = bias only (every score of a row equal: uniform softmax)
    'peaky16_b1_q64': (7, 16.0, 1, 64, 7, (0.0, 1.0)),
    'peaky32_b1_q64': (7, 32.0, 1, 64, 8, (0.0, 1.0)),
    'flat_b1_q64': (7, 0.0, 1, 64, 9, (0.0, 1.0)),
}


def case_inputs(name):
    wseed, gain, b, q, iseed, (lo, hi) = CASES[name]
    sd = synth_state_dict(wseed, attn_gain=gain)
    img, qs = synth_inputs(b, q, iseed)
    qs = qs * (hi - lo) + lo
    return sd, img, qs


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count())
    model = ref_import.build_reference_model()
    only = [a for a in sys.argv[1:] if a in CASES]
    for name in (only or CASES):
        sd, img, qs = case_inputs(name)
        out = {}
        for dt, tag in ((torch.float32, 'f32'), (torch.float64, 'f64')):
            m = model.to(dt)
            m.load_state_dict({k: v.to(dt) for k, v in sd.items()})
            grabbed = {}
            h = m.transformer.register_forward_hook(lambda mod, i, o: grabbed.__setitem__('memory', o[1]))
            pred = m(img.to(dt), qs.to(dt))['pred_corrs']
            h.remove()
            out['pred_' + tag] = pred.numpy()
            mem = grabbed['memory']                      # [B,256,16,32]
            if tag == 'f64':   # fp64 truth, stored rounded to fp32, every 8th token: [B,64,256]
                out['memory'] = mem.flatten(2).permute(0, 2, 1)[:, ::8].contiguous().float().numpy()
        s, s2 = state_checksum(sd)
        out['weights_checksum'] = np.array([s, s2])
        out['meta'] = np.array(list(CASES[name][:5]) + list(CASES[name][5]), dtype=np.float64)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        d = np.abs(out['pred_f32'].astype(np.float64) - out['pred_f64']) * np.array([512.0, 256.0])
        print(f'{name}: pred {out["pred_f32"].shape} fp32-vs-fp64 {d.max():.2e} px -> {path}')


if __name__ == '__main__':
    main()
