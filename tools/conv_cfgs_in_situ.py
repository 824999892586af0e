"""Every distinct convolution of the backbone at B pairs, WITH its real epilogue (FrozenBN, residual where the block has one,
ReLU), under every launch configuration: back-to-back launch time, best few, and the configuration the library picks.
    python tools/conv_cfgs_in_situ.py [B]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
# (name, H, W per half (input), cin, cout, k, stride, residual)
shapes = [('l1 conv1 (blk0)', 64, 64, 64, 64, 1, 1, False), ('l1 conv2', 64, 64, 64, 64, 3, 1, False), ('l1 conv3', 64, 64, 64, 256, 1, 1, True),
          ('l1 conv1', 64, 64, 256, 64, 1, 1, False),
          ('l2 conv1 (blk0)', 64, 64, 256, 128, 1, 1, False), ('l2 conv2 (blk0)', 64, 64, 128, 128, 3, 2, False), ('l2 conv3', 32, 32, 128, 512, 1, 1, True),
          ('l2 conv1', 32, 32, 512, 128, 1, 1, False), ('l2 conv2', 32, 32, 128, 128, 3, 1, False), ('l2 downsample', 64, 64, 256, 512, 1, 2, False),
          ('l3 conv1 (blk0)', 32, 32, 512, 256, 1, 1, False), ('l3 conv2 (blk0)', 32, 32, 256, 256, 3, 2, False), ('l3 conv3', 16, 16, 256, 1024, 1, 1, True),
          ('l3 conv1', 16, 16, 1024, 256, 1, 1, False), ('l3 conv2', 16, 16, 256, 256, 3, 1, False), ('l3 downsample', 32, 32, 512, 1024, 1, 2, False)]
us = ctypes.c_float()
for name, H, W, cin, cout, k, st, has_res in shapes:
    x = torch.randn(B, H, 2 * W, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    Ho, Wo = H // st, W // st
    res = torch.randn(B, Ho, 2 * Wo, cout, device=dev) if has_res else None
    y = torch.empty(B, Ho, 2 * Wo, cout, device=dev)
    out = []
    for cfg in range(lib.cotr_gemm_num_configs()):
        if lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), P(res), 1, P(y), B, H, W, cin, cout, k, st, cfg, sp) != 0:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), P(res), 1, P(y), B, H, W, cin, cout, k, st, cfg, sp)
        e1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 10, cfg))
    pick = lib.cotr_gemm_pick_conv(B, H, W, cin, cout, k, st) if hasattr(lib, 'cotr_gemm_pick_conv') else -1
    tp = dict((c, u) for u, c in out).get(pick, float('nan'))
    out.sort()
    M = B * Ho * 2 * Wo
    print(f'{name:16s} {M:6d}x{cout:4d}x{k * k * cin:4d}  picked cfg{pick} {tp:6.2f} | ' + '  '.join(f'cfg{c} {u:.2f}' for u, c in out[:6]), flush=True)
