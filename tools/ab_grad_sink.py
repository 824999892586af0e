"""Training step with and without the GradSink (deferred one-launch gradient reduction): ms per step, stage 1 and 2.  GPU box.
python tools/ab_grad_sink.py [Q=200]"""
import os, sys, time
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
for stage in (1, 2):
    lrb = 1e-5 if stage == 2 else 0.0
    for use_sink in (False, True):
        m = build_model(cotr_amd.default_args(dropout=0.1, lr_backbone=lrb)).cuda(); m.load_state_dict(synth_state_dict(0)); m.train()
        opt = training.optimizer_for(m, 1e-4, lrb)
        sink = training.grad_sink_for(opt) if use_sink else None
        for _ in range(5): training.train_batch(m, opt, img, q, t, sink=sink)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): training.train_batch(m, opt, img, q, t, sink=sink)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        extra = f'  jobs {sink.last[0]} sources {sink.last[1]}  peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB' if sink else f'  peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB'
        print(f'stage {stage} sink {int(use_sink)}: {dt:7.2f} ms per step{extra}', flush=True)
        del m, opt, sink
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
