"""Pairs per backbone / encoder pass (knob encode_chunk): 32 (default) vs 64 / 128 at large batches.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import _lib
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs
dev = torch.device('cuda', 0)
sd = synth_state_dict(0)
def t(b, q, n, chunk):
    model = build_model(cotr_amd.default_args()).to(dev).eval()    # (a fresh handle per setting: knobs are per handle)
    model.load_state_dict(sd)
    model.set_knob('encode_chunk', chunk)
    img, qs = synth_inputs(b, q, seed=1)
    img, qs = img.to(dev), qs.to(dev)
    for _ in range(2): out = model(img, qs)['pred_corrs']
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): model(img, qs)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
for b, q, n in [(64, 1, 8), (128, 1, 5), (256, 1000, 2), (64, 1000, 4)]:
    res = {}
    ref = None
    for ch in (32, 64, 128):
        if ch > b: continue
        ms, out = t(b, q, n, ch)
        if ref is None: ref = out
        res[ch] = (ms, float((out - ref).abs().max()))
    flop = b * 24.641e9 + b * q * 11.273e6
    print(f'B={b:4d} Q={q:5d}: ' + '  '.join(f'chunk {c}: {ms:8.2f} ms ({flop / ms / 1e9:5.1f} TF, max diff {d:.1e})' for c, (ms, d) in res.items()), flush=True)
