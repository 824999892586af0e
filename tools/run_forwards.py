"""n forwards at (B, Q) and nothing else: the command rocprofv3 counter passes wrap.   python tools/run_forwards.py B Q [n]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

B, Q = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
img, qs = synth_inputs(B, Q, seed=1)
img, qs = img.cuda(), qs.cuda()
for _ in range(n):
    m(img, qs)
torch.cuda.synchronize()
