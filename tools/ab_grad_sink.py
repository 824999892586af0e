"""Training step without / with the GradSink (deferred one-launch gradient reduction) / with GradSink + FusedAdam (one-launch Adam):
ms per step, stage 1 and 2, interleaved repetitions (min and median of 5 x 20 steps per variant).  GPU box.
python tools/ab_grad_sink.py [Q=200]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
B, Q = 16, (int(sys.argv[1]) if len(sys.argv) > 1 else 200)
g = torch.Generator().manual_seed(0)
img = torch.randn(B, 3, 256, 512, generator=g).cuda()
q, t = torch.rand(B, Q, 2, generator=g).cuda(), torch.rand(B, Q, 2, generator=g).cuda()
NAMES = ['per-weight reductions, torch Adam', 'GradSink, torch Adam', 'GradSink + FusedAdam']
for stage in (1, 2):
    lrb = 1e-5 if stage == 2 else 0.0
    variants = []
    for mode in (0, 1, 2):
        m = build_model(cotr_amd.default_args(dropout=0.1, lr_backbone=lrb)).cuda(); m.load_state_dict(synth_state_dict(0)); m.train()
        opt = training.optimizer_for(m, 1e-4, lrb, fused=(mode == 2))
        sink = training.grad_sink_for(opt) if mode > 0 else None
        for _ in range(5): training.train_batch(m, opt, img, q, t, sink=sink)
        variants.append((m, opt, sink, []))
    for rep in range(5):
        for m, opt, sink, times in variants:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): training.train_batch(m, opt, img, q, t, sink=sink)
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 20 * 1e3)
    for name, (_, _, sink, times) in zip(NAMES, variants):
        extra = f'  ({sink.last[0]} jobs, {sink.last[1]} sources per flush)' if sink is not None else ''
        print(f'stage {stage}  {name:36s} min {min(times):6.2f}  median {statistics.median(times):6.2f} ms per step{extra}', flush=True)
    del variants
    torch.cuda.empty_cache()
