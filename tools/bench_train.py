"""Stage-1 training step time (cotr_amd/training.py) at the reference's batch shapes. GPU box.
    python tools/bench_train.py [pairs] [queries]      (train_cotr.py defaults: --batch_size 32/24/16, --num_kp 100)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m = build_model(cotr_amd.default_args()).cuda()
m.load_state_dict(synth_state_dict(0))
m.train()
opt = training.optimizer_for(m)
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(B, Q, 2, generator=g).cuda(), torch.rand(B, Q, 2, generator=g).cuda()
for _ in range(3):
    training.train_batch(m, opt, img, q, t)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    loss, _ = training.train_batch(m, opt, img, q, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    training.backbone_features(m, img)
torch.cuda.synchronize()
db = (time.perf_counter() - t0) / n
print(f'stage-1 train step, {B} pairs x {Q} queries, cycle + bidirectional: {dt * 1e3:.1f} ms/step = {B / dt:.0f} pairs/s '
      f'(frozen backbone on the HIP kernels: {db * 1e3:.1f} ms of it), loss {loss:.4f}', flush=True)
