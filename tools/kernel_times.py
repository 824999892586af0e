"""In-situ per-launch HIP-event timing of one forward (no rocprof overhead). GPU box.
    python tools/kernel_times.py [B] [Q] [KNOB=INT ...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
for kv in sys.argv[3:]:
    m.set_knob(kv.split('=')[0], int(kv.split('=')[1]))
img, qs = synth_inputs(B, Q, seed=1)
img, qs = img.cuda(), qs.cuda()
for _ in range(30):
    m(img, qs)
torch.cuda.synchronize()
m.set_profiling(2)
acc = collections.OrderedDict()
N = 10
for it in range(N):
    m(img, qs)
    torch.cuda.synchronize()
    for i, (n, t) in enumerate(m.get_profile()):
        k = (i, n)
        acc[k] = acc.get(k, 0.0) + t
m.set_profiling(0)
tot = 0
fam = collections.defaultdict(float)
for (i, n), t in acc.items():
    us = t / N * 1e3
    tot += us
    fam[n.split(' ')[0]] += us
    print(f'{i:3d} {n:44s} {us:8.2f} us')
print(f'total {tot:.1f} us over {len(acc)} launches')
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f'   {k:20s} {v:8.1f} us')
