"""GEMM shapes of one training step (forward + dX GEMMs of cotr_amd/train_ops.py), the configuration the library picks for each
and the best one measured: candidates for gemm_tuned.inc.   python tools/tune_train_shapes.py [stage=1|2]   (GPU box)"""
import collections
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import _lib, training, train_ops
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict

lib = _lib.load_library()
B, Q = 16, 200   # bench.py --workload train: BASELINE.json configs[4]
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LRB = 1e-5 if STAGE == 2 else 0.0
m = build_model(cotr_amd.default_args(lr_backbone=LRB)).cuda()
m.load_state_dict(synth_state_dict(0))
m.train()
opt = training.optimizer_for(m, 1e-4, LRB)
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(B, Q, 2, generator=g).cuda(), torch.rand(B, Q, 2, generator=g).cuda()
training.train_batch(m, opt, img, q, t)
shapes = collections.Counter()
orig = train_ops.gemm


def rec(a, w, bias=None, relu=False):
    shapes[(a.shape[0], w.shape[0], a.shape[1])] += 1
    return orig(a, w, bias, relu)


train_ops.gemm = rec
training.train_batch(m, opt, img, q, t)
train_ops.gemm = orig
torch.cuda.synchronize()
P = lambda x: x.data_ptr()
ncfg = lib.cotr_gemm_num_configs()
us = ctypes.c_float()
tot_pick = tot_best = 0.0
for (M, N, K), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda')
    y = torch.empty(M, N, device='cuda')
    res = {}
    for cfg in [-1] + list(range(ncfg)):
        rc = lib.cotr_bench_linear(P(x), P(w), None, P(y), M, N, K, cfg, 20, ctypes.byref(us))
        if rc == 0:
            res[cfg] = us.value
    best = min((v, k) for k, v in res.items() if k >= 0)
    pick = res.get(-1, float('nan'))
    tot_pick += cnt * pick
    tot_best += cnt * best[0]
    print(f'{M:6d} x {N:5d} x {K:5d}  x{cnt:3d}/step   picked {pick:7.2f} us   best cfg {best[1]:2d} {best[0]:7.2f} us   '
          f'{{0, {M}, {N}, {K}, {best[1]}, {best[1]}}},', flush=True)
print(f'per step: picked {tot_pick / 1e3:.2f} ms, best {tot_best / 1e3:.2f} ms')
