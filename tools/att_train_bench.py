"""Training attention kernels per shape, kernel time from the torch profiler (the calls are host-bound for the small shapes):
forward, delta + dQ, dK/dV per launch, for both forms of the backward kernels (cotr_set_train_attention_form).  GPU box.
python tools/att_train_bench.py"""
import os, sys, collections, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cotr_amd import train_ops as T, _lib

for form in (1, 2, 3, 0):
    _lib.set_knob('train_attention_form', form)
    for name, nb, nq, p in (('encoder 32 x 512', 32, 512, 0.1), ('decoder 16 x 200', 16, 200, 0.1), ('decoder 16 x 200 p=0', 16, 200, 0.0), ('decoder 8 x 200', 8, 200, 0.1)):
        g = torch.Generator().manual_seed(0)
        q = torch.randn(nb * nq, 256, generator=g).cuda().requires_grad_()
        k = torch.randn(nb * 512, 256, generator=g).cuda().requires_grad_()
        v = torch.randn(nb * 512, 256, generator=g).cuda().requires_grad_()
        do = torch.randn(nb * nq, 256, generator=g).cuda()
        gf = 4.0 * nb * nq * 512 * 256 / 1e9          # forward: 2 products
        def run():
            o = T.Attention.apply(None, q, k, v, nb, nq, 32 ** -0.5, p)
            torch.autograd.grad(o, (q, k, v), do)
        for _ in range(3): run()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10): run()
            torch.cuda.synchronize()
        us = collections.defaultdict(float)
        for e in prof.key_averages():
            mm = re.search(r'attn_\w+', e.key)
            if mm:
                us[mm.group(0)] += e.device_time_total / 10
            elif 'sum_parts' in e.key:
                us['sum_parts'] += e.device_time_total / 10
        parts = '  '.join(f'{k_} {v_:7.1f} us' for k_, v_ in sorted(us.items()))
        fwd = us.get('attn_train_fwd_kernel', float('nan'))
        bwd = sum(v_ for k_, v_ in us.items() if 'bwd' in k_ or 'delta' in k_ or 'sum_parts' in k_)
        print(f'form {form}  {name:22s} {parts}   | forward {gf / fwd * 1e3:5.1f} TFLOP/s, backward {2.5 * gf / bwd * 1e3:5.1f} TFLOP/s of the 5 products needed', flush=True)
