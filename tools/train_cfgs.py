"""Round 6: the Linear shapes of one training step (forward + dX GEMMs of cotr_amd/train_ops.py, recorded from a real step) under every
launch configuration, ROUND-ROBIN against the library's pick (tools/mid_batch_cfgs.py says why) - candidates for csrc/gemm_tuned.inc in
the format tools/apply_mid_batch_cfgs.py reads, and what the step's GEMMs cost under the pick / the best.
    python tools/train_cfgs.py [stage=1|2]"""
import collections
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
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
    shapes[(a.shape[0], w.shape[0], a.shape[1], bool(relu), bias is not None)] += 1
    return orig(a, w, bias, relu)


train_ops.gemm = rec
training.train_batch(m, opt, img, q, t)
train_ops.gemm = orig
torch.cuda.synchronize()
P = lambda x: None if x is None else x.data_ptr()
sp = _lib.current_stream_ptr()
NCFG = lib.cotr_gemm_num_configs()
REPS = 30


def measure(cands):
    live = [(lab, fn) for lab, fn in cands if fn() == 0]
    torch.cuda.synchronize()
    for _ in range(3 * REPS):
        live[0][1]()
    best = {lab: 1e9 for lab, _ in live}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        for lab, fn in live:
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best[lab] = min(best[lab], e0.elapsed_time(e1) * 1000.0 / REPS)
    return best


tot_pick = tot_best = 0.0
merged = collections.Counter()
for (M, N, K, relu, has_b), cnt in shapes.items():
    merged[(M, N, K)] += cnt
for (M, N, K), cnt in sorted(merged.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
    x, w, b = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') / K ** 0.5, torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    cands = [('pick', lambda: lib.cotr_op_linear(P(x), None, 0, P(w), None, P(b), None, 0, P(y), M, N, K, sp))]
    cands += [(cfg, (lambda c: lambda: lib.cotr_op_linear_cfg(P(x), P(w), P(b), None, 0, P(y), M, N, K, c, sp))(cfg)) for cfg in range(NCFG)]
    res = measure(cands)
    pick = res['pick']
    out = sorted((u, c) for c, u in res.items() if c != 'pick')
    best_u, best_c = out[0]
    reg = [(u, c) for u, c in out if c <= 18]
    reg_c = reg[0][1] if reg else best_c
    tot_pick += cnt * pick
    tot_best += cnt * min(best_u, pick)
    flag = f'   <== {{0, {M}, {N}, {K}, {best_c}, {reg_c}}},  // {best_u:.2f} us (pick {pick:.2f})' if best_u < 0.97 * pick else ''
    print(f'{M:6d} x {N:5d} x {K:5d}  x{cnt:3d}/step  pick {pick:7.2f} | ' + '  '.join(f'cfg{c} {u:.2f}' for u, c in out[:4]) + flag, flush=True)
print(f'# the step\'s Linear / dX GEMMs: {tot_pick / 1e3:.2f} ms under the picks, {tot_best / 1e3:.2f} ms under the best of each shape')
