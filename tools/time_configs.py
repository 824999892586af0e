"""Forward time at the shapes BASELINE.json's configs name (and the engine's), HIP events around n calls. GPU box.
    python tools/time_configs.py [KNOB=INT ...]      (knobs of the model's handle, e.g. split_f16=1 with COTR_HIP_EXPERIMENTAL=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
for kv in sys.argv[1:]:
    m.set_knob(kv.split('=')[0], int(kv.split('=')[1]))
if sys.argv[1:]:
    print('knobs:', ' '.join(sys.argv[1:]))
for tag, B, Q, n in (('configs[1] primary', 1, 1000, 200), ('engine batch (sparse_engine.py:47-56)', 32, 1, 30),
                     ('dense pass, one pair (inference_helper.py:116-127)', 1, 131072, 10),
                     ('dense pass, 2x2 patch pairs', 4, 131072, 5), ('32 pairs x 1000', 32, 1000, 10),
                     ('configs[3] batch', 256, 1000, 3)):
    img, qs = synth_inputs(B, Q, seed=1)
    img, qs = img.cuda(), qs.cuda()
    for _ in range(3):
        m(img, qs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        m(img, qs)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    flop = B * 24.641e9 + B * Q * 11.273e6
    print(f'{tag:52s} B={B:4d} Q={Q:7d}: {ms:9.3f} ms  {B * Q / ms * 1e3:12.0f} query-corr/s  {B / ms * 1e3:8.1f} pairs/s  '
          f'{flop / ms / 1e9:6.1f} TFLOP/s = {flop / ms / 1e9 / 157.3 * 100:4.1f} % of fp32 MFMA peak', flush=True)
