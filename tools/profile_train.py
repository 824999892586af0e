"""torch.profiler table of one stage-1 training step. GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
B, Q = 16, 100
m = build_model(cotr_amd.default_args()).cuda(); m.load_state_dict(synth_state_dict(0)); m.train()
opt = training.optimizer_for(m)
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(B, Q, 2, generator=g).cuda(), torch.rand(B, Q, 2, generator=g).cuda()
for _ in range(3): training.train_batch(m, opt, img, q, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): training.train_batch(m, opt, img, q, t)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=35, max_name_column_width=60))
