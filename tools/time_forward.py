"""Time model forward at (B,Q) for attention split settings; prints ms per forward. GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import _lib
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

def main():
    m = build_model(cotr_amd.default_args()).cuda().eval()
    m.load_state_dict(synth_state_dict(0))
    lib = _lib.load_library()
    shapes = [(1, 1000), (2, 257), (4, 1000), (8, 1000), (32, 1), (32, 1000)]
    shapes = [(1, 1000), (2, 257), (3, 500), (4, 1000), (5, 100), (8, 1000), (13, 1), (16, 1), (32, 1), (32, 1000)]
    shapes = [(1, 1000), (1, 4000), (2, 257), (4, 1000), (32, 1000)]
    shapes = [(1, 1000), (1, 2500), (2, 257), (4, 500), (6, 400), (8, 1000), (32, 1)]
    for ns in [0, 3072]:
        m.set_knob('ffn_fusion_max_rows', ns)
        for (b, q) in shapes:
            img, qs = synth_inputs(b, q, seed=1)
            img, qs = img.cuda(), qs.cuda()
            for _ in range(5): m(img, qs)
            torch.cuda.synchronize()
            n = 30
            t = time.perf_counter()
            for _ in range(n): m(img, qs)
            torch.cuda.synchronize()
            print(f'ffn_fuse_max_rows={ns} B={b} Q={q}: {(time.perf_counter()-t)/n*1e3:.3f} ms', flush=True)
main()
