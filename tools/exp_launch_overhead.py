"""Launch-side experiments at the metric point (1 pair x 1000 queries): eager launches vs one captured HIP graph, under
whatever HIP runtime environment the caller set (tools/exp_launch_overhead.sh sweeps HIP_FORCE_DEV_KERNARG).
    python tools/exp_launch_overhead.py [eager|graph] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda', 0)
model = build_model(cotr_amd.default_args()).to(dev).eval()
model.load_state_dict(synth_state_dict(0))
img, qs = synth_inputs(1, 1000, seed=1)
img, qs = img.to(dev), qs.to(dev)
model.reserve(1, 1000)
for _ in range(20):
    ref = model(img, qs)['pred_corrs']
torch.cuda.synchronize()
if mode == 'graph':
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model(img, qs)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = model(img, qs)['pred_corrs']
    torch.cuda.synchronize()
    fn = g.replay
    g.replay()
    torch.cuda.synchronize()
    print('graph output == eager output:', bool(torch.equal(out, ref)))
else:
    fn = lambda: model(img, qs)
for _ in range(20):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / steps)
print(f'{mode:6s} HIP_FORCE_DEV_KERNARG={os.environ.get("HIP_FORCE_DEV_KERNARG", "unset")}: {best * 1e3:.1f} us per forward', flush=True)
