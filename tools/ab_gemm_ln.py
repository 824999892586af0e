"""A 256-wide projection + LayerNorm: large-tile GEMM then layernorm_kernel (two launches) against gemm_ln_kernel (one), per shape.
GPU box.   python tools/ab_gemm_ln.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else t.data_ptr()
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, K in ((32768, 256), (32768, 1024), (32000, 256), (32000, 1024), (65536, 256)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(256, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(256, generator=g).cuda(); r = torch.randn(M, 256, generator=g).cuda()
    lw, lb = torch.rand(256, generator=g).cuda() + 0.5, torch.randn(256, generator=g).cuda()
    tmp, y = torch.empty(M, 256, device='cuda'), torch.empty(M, 256, device='cuda')
    sp = _lib.current_stream_ptr()
    def two():
        lib.cotr_op_linear(P(x), None, 0, P(w), None, P(b), P(r), 0, P(tmp), M, 256, K, sp)
        lib.cotr_op_layernorm(P(tmp), P(lw), P(lb), P(y), M, sp)
    one = lambda: lib.cotr_op_linear_ln(P(x), P(w), P(b), P(r), P(lw), P(lb), P(y), M, K, sp)
    t2, t1 = timed(two), timed(one)
    gf = 2.0 * M * 256 * K / 1e9
    print(f'{M:6d} x 256 x {K:4d}: GEMM + layernorm {t2:7.1f} us   one launch {t1:7.1f} us ({gf / t1 * 1e3:5.1f} TFLOP/s)', flush=True)
