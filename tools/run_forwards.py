"""n forwards at (B, Q) and nothing else: the command rocprofv3 counter passes wrap.   python tools/run_forwards.py B Q [n] [KNOB=INT ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

B, Q = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
knobs = [a for a in sys.argv[4:] if '=' in a]
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
for kv in knobs:
    m.set_knob(kv.split('=')[0], int(kv.split('=')[1]))
img, qs = synth_inputs(B, Q, seed=1)
img, qs = img.cuda(), qs.cuda()
for _ in range(n):
    m(img, qs)
torch.cuda.synchronize()
