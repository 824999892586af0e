"""cProfile of the HOST side of eager training steps (16 pairs x 200 queries, stage 1, GradSink + FusedAdam): where the ~16 ms of Python
per step go.  GPU box.   python tools/host_profile_train.py [stage]"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LRB = 1e-5 if STAGE == 2 else 0.0
m = build_model(cotr_amd.default_args(dropout=0.1, lr_backbone=LRB)).cuda(); m.load_state_dict(synth_state_dict(0)); m.train()
opt = training.optimizer_for(m, 1e-4, LRB, fused=True)
g = torch.Generator().manual_seed(0)
img = torch.randn(16, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(16, 200, 2, generator=g).cuda(), torch.rand(16, 200, 2, generator=g).cuda()
for _ in range(5): training.train_batch(m, opt, img, q, t)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): training.train_batch(m, opt, img, q, t)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(28)
print(out.getvalue())
