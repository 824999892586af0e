"""Where the time of ONE small convolution launch goes: phase timestamps written by every workgroup of the k-split kernels
(cotr_debug_conv_times: entry, loads issued, first data usable, K loop done, stored; 100 MHz wall clock = 10 ns ticks).
    python tools/conv_phases.py          (GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib

lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
# (name, B, H, W(per half), Cin, Cout, k, stride, configs)
shapes = [('layer3 conv2 3x3', 1, 16, 16, 256, 256, 3, 1, (24, 30, 31, 33, 39, 32)),
          ('layer3 conv1 1x1 1024->256', 1, 16, 16, 1024, 256, 1, 1, (24, 30, 33, 39, 32)),
          ('layer3 conv3 1x1 256->1024', 1, 16, 16, 256, 1024, 1, 1, (4, 14, 34, 36, 32)),
          ('layer2 conv2 3x3', 1, 32, 32, 128, 128, 3, 1, (19, 3, 13, 32, 34, 38)),
          ('layer2 conv1 1x1 512->128', 1, 32, 32, 512, 128, 1, 1, (19, 32, 34, 38)),
          ('layer1 conv2 3x3', 1, 64, 64, 64, 64, 3, 1, (14, 34, 38, 32, 35, 37)),
          ('layer1 conv1 1x1 256->64', 1, 64, 64, 256, 64, 1, 1, (10, 35, 32, 34))]
for name, B, H, W, cin, cout, k, st, cfgs in shapes:
    x = torch.randn(B, H, 2 * W, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(B, H // st, 2 * (W // st), cout, device=dev)
    for cfg in cfgs:
        times = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
        ok = True
        for it in range(6):      # the last launch is the one read back (caches warm, clocks up)
            times.zero_()
            torch.cuda.synchronize()
            rc = lib.cotr_debug_conv_times(P(x), P(w), P(sc), P(bi), P(y), B, H, W, cin, cout, k, st, cfg, P(times), sp)
            if rc != 0:
                ok = False
                break
        torch.cuda.synchronize()
        if not ok:
            print(f'{name:28s} cfg {cfg:2d}: declined')
            continue
        t = times.cpu()
        used = t[:, 0] > 0
        t = t[used].double()
        t0 = t[:, 0].min()
        rel = (t - t0) * 0.01           # us since the first workgroup entered
        names = ['entry', 'loads issued', 'data usable', 'K loop done', 'stored']
        line = '  '.join(f'{n} {rel[:, i].mean():5.2f} (max {rel[:, i].max():5.2f})' for i, n in enumerate(names))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, H, W, cin, cout, k, st, cfg, sp)
        e1.record()
        torch.cuda.synchronize()
        print(f'{name:28s} cfg {cfg:2d}: {int(used.sum()):4d} wgs  {e0.elapsed_time(e1) * 20:6.2f} us/launch back-to-back | us since first entry: {line}')
