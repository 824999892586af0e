"""1x1 expansions (conv3 + FrozenBN + residual + ReLU) of one pair under every launch configuration, back-to-back launch time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib

lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
for name, B, H, W, cin, cout in [('layer1 conv3', 1, 64, 64, 64, 256), ('layer2 conv3', 1, 32, 32, 128, 512), ('layer3 conv3', 1, 16, 16, 256, 1024),
                                 ('layer1 conv1', 1, 64, 64, 256, 64), ('layer2 conv1', 1, 32, 32, 512, 128)]:
    x = torch.randn(B, H, 2 * W, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    sc, bi = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    res = torch.randn(B, H, 2 * W, cout, device=dev) if 'conv3' in name else None
    y = torch.empty(B, H, 2 * W, cout, device=dev)
    out = []
    for cfg in range(lib.cotr_gemm_num_configs()):
        if lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), P(res), 1, P(y), B, H, W, cin, cout, 1, 1, cfg, sp) != 0:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), P(res), 1, P(y), B, H, W, cin, cout, 1, 1, cfg, sp)
        e1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 10, cfg))
    out.sort()
    print(f'{name:14s} M={B * H * 2 * W:5d} {cin:4d}->{cout:4d}: ' + '  '.join(f'cfg{c} {u:.2f}' for u, c in out[:8]), flush=True)
