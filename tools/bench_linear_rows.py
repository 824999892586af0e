"""A/B of the K-short products for many rows: experimental/linear_rows.hip (A tile resident, weights streamed; research library, off by
default) against the tuned tile kernels on the shapes of the batched forward.  Lost: profiles/r5_ab_linear_rows_lost.txt.
    COTR_HIP_EXPERIMENTAL=1 python tools/bench_linear_rows.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotr_amd import _lib  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402
from tools.bench_ffn_rows import timeit  # noqa: E402


def main():
    d = G.dev()
    g = torch.Generator().manual_seed(0)
    print('# M x N x K (what) | rows kernel us (of 157.3 TFLOP/s) | tile kernels us | max rel diff')
    for M, N, K, res, relu, what in ((16384, 768, 256, True, False, 'encoder in-projection, 32 pairs'), (16384, 3072, 256, True, False, 'decoder K/V projection'),
                                    (32000, 256, 256, False, True, 'corr_embed layer'), (65536, 512, 128, True, True, 'layer2 conv3 1x1, 32 pairs'),
                                    (16384, 1024, 256, True, True, 'layer3 conv3 1x1, 32 pairs'), (32768, 768, 256, True, False, 'encoder in-projection, 64 pairs')):
        x = torch.randn(M, K, generator=g).to(d)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(d)
        b = (torch.randn(N, generator=g) * 0.1).to(d)
        r = torch.randn(M, N, generator=g).to(d) if res else None
        run = lambda: G.op_linear(x, w, bias=b, residual=r, relu=relu)
        y0 = run()
        t0 = timeit(run)
        try:
            _lib.set_knob('linear_rows_min_rows', 8192)
            y = run()
            t1 = timeit(run)
        finally:
            _lib.reset_knobs()
        fl = 2.0 * M * N * K
        print(f'{M:6d} x {N:4d} x {K:3d} ({what:32s}) | {t1:7.1f} ({fl / t1 * 1e-6 / 157.3:.3f}) | {t0:7.1f} ({fl / t0 * 1e-6 / 157.3:.3f}) | {G.rel_err(y, y0):.1e}')


if __name__ == '__main__':
    main()
