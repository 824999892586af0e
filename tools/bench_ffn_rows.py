"""A/B of the FFN block for many rows: ONE launch (ffn_rows.hip) against linear1 + linear2 (+residual) + LayerNorm on the tuned GEMM
configurations (the path it replaces).  Prints microseconds per block and TFLOP/s of the fp32-MFMA peak (157.3).

    python tools/bench_ffn_rows.py [rows ...]          -> profiles/r5_ab_ffn_rows.txt is its output on the MI355X
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotr_amd import _lib  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    rows = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 20000, 32000, 32768, 65536]
    lib = _lib.load_library()
    d = G.dev()
    g = torch.Generator().manual_seed(0)
    w1, b1 = (torch.randn(1024, 256, generator=g) / 16).to(d), (torch.randn(1024, generator=g) * 0.1).to(d)
    w2, b2 = (torch.randn(256, 1024, generator=g) / 32).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
    lw, lb = (torch.rand(256, generator=g) + 0.5).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
    print('# rows | one launch us (TFLOP/s, of peak) | three launches us (TFLOP/s, of peak) | max rel diff')
    for M in rows:
        x = torch.randn(M, 256, generator=g).to(d)
        y = torch.empty(M, 256, device=d)
        hid = torch.empty(M, 1024, device=d)
        tmp = torch.empty(M, 256, device=d)
        y3 = torch.empty(M, 256, device=d)
        s = G.sptr()

        def one():
            assert lib.cotr_op_ffn_rows(G.P(x), G.P(w1), G.P(b1), G.P(w2), G.P(b2), G.P(lw), G.P(lb), None, None, G.P(y), M, s) == 0

        def three():
            assert lib.cotr_op_linear(G.P(x), None, 0, G.P(w1), None, G.P(b1), None, 1, G.P(hid), M, 1024, 256, s) == 0
            assert lib.cotr_op_linear(G.P(hid), None, 0, G.P(w2), None, G.P(b2), G.P(x), 0, G.P(tmp), M, 256, 1024, s) == 0
            assert lib.cotr_op_layernorm(G.P(tmp), G.P(lw), G.P(lb), G.P(y3), M, s) == 0

        t1, t3 = timeit(one), timeit(three)
        fl = 2 * 2 * 256 * 1024 * M
        diff = G.rel_err(y, y3)
        print(f'{M:6d} | {t1:8.1f} ({fl / t1 * 1e-6:6.1f}, {fl / t1 * 1e-6 / 157.3:.3f}) | {t3:8.1f} ({fl / t3 * 1e-6:6.1f}, {fl / t3 * 1e-6 / 157.3:.3f}) | {diff:.2e}')


if __name__ == '__main__':
    main()
