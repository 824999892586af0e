"""A/B of the fused layer1 bottleneck (knob bottleneck_max_pairs) across batch sizes: ms per forward with it off / on.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import _lib
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs
dev = torch.device('cuda', 0)
model = build_model(cotr_amd.default_args()).to(dev).eval()
model.load_state_dict(synth_state_dict(0))
def t(b, q, n):
    img, qs = synth_inputs(b, q, seed=1)
    img, qs = img.to(dev), qs.to(dev)
    for _ in range(3): model(img, qs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): model(img, qs)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for b, q, n in [(1, 1000, 100), (2, 1000, 50), (4, 1000, 30), (8, 1000, 20), (16, 1, 10), (32, 1, 10), (32, 1000, 5)]:
    model.set_knob('bottleneck_max_pairs', 0); off = t(b, q, n)
    model.set_knob('bottleneck_max_pairs', 64); on = t(b, q, n)
    print(f'B={b:3d} Q={q:5d}: unfused {off:8.3f} ms  fused {on:8.3f} ms  ({100 * (on - off) / off:+.1f} %)', flush=True)
model.reset_knobs()
