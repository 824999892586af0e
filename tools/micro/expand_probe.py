"""K-short, residual-carrying GEMMs (the 1x1 expansions of the bottlenecks at 32 pairs): time per config with and
without the residual, torch events around 20 stream launches. GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for M, N, K in ((262144, 256, 64), (65536, 512, 128), (16384, 1024, 256), (262144, 64, 256), (16384, 256, 2304)):
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / 8; b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda'); y = torch.empty(M, N, device='cuda')
    for res in (None, r):
        row = []
        for c in (0, 1, 2, 26, 27):
            sp = _lib.current_stream_ptr()
            if lib.cotr_op_linear_cfg(P(x), P(w), P(b), P(res), 1, P(y), M, N, K, c, sp) != 0:
                row.append(float('nan')); continue
            row.append(timeit(lambda: lib.cotr_op_linear_cfg(P(x), P(w), P(b), P(res), 1, P(y), M, N, K, c, sp)))
        mb = (M * N * (2 if res is not None else 1) + M * K) * 4 / 1e6
        best = min(t for t in row if t == t)
        print(f'M={M:7d} N={N:5d} K={K:4d} res={int(res is not None)}  ' + '  '.join(f'{t:7.1f}' for t in row) +
              f'  us (cfg 0 1 2 26 27)  {mb / best:5.2f} TB/s  {2.0 * M * N * K / best / 1e6:6.1f} TFLOP/s', flush=True)
