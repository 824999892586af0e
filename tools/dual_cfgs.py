"""Entry blocks of layer2 / layer3 at one pair: the downsample conv + conv1 of the same input as ONE launch (conv_pair, api.hip) under
every configuration that has a dual form, against the two launches one by one.   python tools/dual_cfgs.py   (GPU box)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, H, cin, c_ds, c_1, stride in [('layer2 entry', 64, 256, 512, 128, 2), ('layer3 entry', 32, 512, 1024, 256, 2)]:
    x = torch.randn(1, H, 2 * H, cin, device=dev)
    wd, w1 = torch.randn(c_ds, cin, device=dev) / cin ** 0.5, torch.randn(c_1, cin, device=dev) / cin ** 0.5
    sd, bd, s1, b1 = torch.ones(c_ds, device=dev), torch.zeros(c_ds, device=dev), torch.ones(c_1, device=dev), torch.zeros(c_1, device=dev)
    yd = torch.empty(1, H // stride, 2 * H // stride, c_ds, device=dev)
    y1 = torch.empty(1, H, 2 * H, c_1, device=dev)
    row = []
    for cfg in range(lib.cotr_gemm_num_configs()):
        call = lambda: lib.cotr_op_conv_dual_cfg(P(x), P(wd), P(sd), P(bd), 0, P(yd), c_ds, 1, stride, P(w1), P(s1), P(b1), 1, P(y1), c_1, 1, 1, 1, H, H, cin, cfg, sp)
        if call() != 0:
            continue
        row.append((timeit(call), cfg))
    sep = []
    for cfg in range(lib.cotr_gemm_num_configs()):
        cd = lambda: lib.cotr_op_conv_cfg(P(x), P(wd), P(sd), P(bd), None, 0, P(yd), 1, H, H, cin, c_ds, 1, stride, cfg, sp)
        c1 = lambda: lib.cotr_op_conv_cfg(P(x), P(w1), P(s1), P(b1), None, 1, P(y1), 1, H, H, cin, c_1, 1, 1, cfg, sp)
        td = timeit(cd) if cd() == 0 else None
        t1 = timeit(c1) if c1() == 0 else None
        sep.append((cfg, td, t1))
    bd_ = min((t, c) for c, t, _ in sep if t); b1_ = min((t, c) for c, _, t in sep if t)
    row.sort()
    print(f'{name}: dual ' + '  '.join(f'cfg{c} {t:.2f}' for t, c in row[:8]) + f' | separate best: downsample cfg{bd_[1]} {bd_[0]:.2f} + conv1 cfg{b1_[1]} {b1_[0]:.2f} = {bd_[0] + b1_[0]:.2f}', flush=True)
