"""torch.profiler table of training steps: python tools/profile_train.py [stage=1|2] [Q=200] [sink=0|1|2 (2 = GradSink + FusedAdam)]. GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, Q = 16, (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
LRB = 1e-5 if STAGE == 2 else 0.0
m = build_model(cotr_amd.default_args(dropout=0.1, lr_backbone=LRB)).cuda(); m.load_state_dict(synth_state_dict(0)); m.train()
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
opt = training.optimizer_for(m, 1e-4, LRB, fused=(MODE == 2))
SINK = training.grad_sink_for(opt) if MODE else None
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(B, Q, 2, generator=g).cuda(), torch.rand(B, Q, 2, generator=g).cuda()
for _ in range(3): training.train_batch(m, opt, img, q, t, sink=SINK)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): training.train_batch(m, opt, img, q, t, sink=SINK)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60))
