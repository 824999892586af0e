"""Attention kernel variants at many query rows: time per launch (HIP events), TFLOP/s, bit-equality against the 32-query kernel.
usage: python tools/att_bench.py [--variant narrow|wide2|wide3] [--iters N]   (one variant only: for rocprofv3 --pmc runs)"""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument('--variant', default=None)
ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: t.data_ptr()
sptr = _lib.current_stream_ptr()


def setv(name):
    if name == 'narrow':
        _lib.set_knob('attention_wide_min_rows', 1 << 30)
    else:
        _lib.set_knob('attention_wide_min_rows', 0)
        _lib.set_knob('attention_wide_occupancy', int(name[4:]))


shapes = [(32, 1000), (32, 512), (4, 8192), (1, 32768), (3, 77)]
variants = [args.variant] if args.variant else ['narrow', 'wide2', 'wide3']
g = torch.Generator().manual_seed(1)
for nb, nq in shapes:
    R = nb * nq
    q = torch.randn(R, 256, generator=g).to(dev)
    kv = torch.randn(nb * 512, 512, generator=g).to(dev)
    outs = {}
    for vn in variants:
        setv(vn)
        o = torch.full((R, 256), float('nan'), device=dev)
        run = lambda: lib.cotr_op_attention(P(q), 256, P(kv), P(kv[:, 256:]), 512, P(o), 256, nb, nq, sptr)
        assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            run()
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = R * 512 * 256 * 4.0
        outs[vn] = o.clone()
        same = '' if vn == 'narrow' or 'narrow' not in outs else ('  bit-identical' if torch.equal(outs[vn], outs['narrow']) else
                                                                  '  DIFFERS max %.3e' % (outs[vn] - outs['narrow']).abs().max().item())
        print('nb %3d nq %6d  %-8s %9.2f us  %6.1f TFLOP/s  (%.3f of 157.3)%s' % (nb, nq, vn, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3, same),
              flush=True)
_lib.reset_knobs()
