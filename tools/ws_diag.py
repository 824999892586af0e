"""Where the K loop of the wave-specialised large-tile GEMM (configs 40 / 41) spends its cycles: barrier waits vs the rest, and the
shader clock it runs at (cotr_debug_conv_times -> GemmParams::dbg, written by MFMA wavefront 0 of every workgroup).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
for name, B, H, W, cin, cout, k, st in [('l3 conv2 3x3 16384x256x2304', 32, 16, 16, 256, 256, 3, 1), ('l3 conv1 1x1 16384x256x1024', 32, 16, 16, 1024, 256, 1, 1),
                                         ('l2 conv2 3x3 65536x128x1152', 32, 32, 32, 128, 128, 3, 1), ('l3 conv3 1x1 16384x1024x256', 32, 16, 16, 256, 1024, 1, 1)]:
    x = torch.randn(B, H, 2 * W, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(B, H // st, 2 * (W // st), cout, device=dev)
    for cfg, flags in ((26, 0), (27, 0), (40, 0), (40, 2), (41, 0), (41, 2)):
        _lib.set_knob('ws_flags', flags)
        if lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, H, W, cin, cout, k, st, cfg, sp) != 0:
            print(name, cfg, 'declined'); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, H, W, cin, cout, k, st, cfg, sp)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        M, N, K = y.numel() // cout, cout, k * k * cin
        print(f'{name:30s} cfg {cfg} flags {flags}: {us:7.1f} us/launch {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s', flush=True)
