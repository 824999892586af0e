"""A/B of layer1 block 0's two 1x1 convolutions over the pooled stem output for many pairs: downsample (64 -> 256) + conv1 (64 -> 64, ReLU)
in ONE launch (expand.hip) against the two tuned GEMM launches.  Prints microseconds and the bytes moved per second (x once, both outputs).

    python tools/bench_expand.py [pairs ...]          -> profiles/r5_ab_expand.txt is its output on the MI355X
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotr_amd import _lib  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402
from tools.bench_ffn_rows import timeit  # noqa: E402


def main():
    pairs = [int(a) for a in sys.argv[1:]] or [5, 8, 16, 32, 64]
    lib = _lib.load_library()
    d = G.dev()
    g = torch.Generator().manual_seed(0)
    mk = lambda n, k: ((torch.randn(n, k, generator=g) / math.sqrt(k)).to(d), (torch.rand(n, generator=g) + 0.5).to(d), torch.randn(n, generator=g).to(d))
    wd, sd, bd = mk(256, 64)
    w1, s1, b1 = mk(64, 64)
    print('# pairs | one launch us (TB/s) | two launches us | max rel diff')
    for B in pairs:
        s = G.sptr()
        M1 = B * 64 * 128
        x = torch.randn(M1, 64, generator=g).to(d)
        yd, y1 = torch.empty(M1, 256, device=d), torch.empty(M1, 64, device=d)
        yd2, y12 = torch.empty(M1, 256, device=d), torch.empty(M1, 64, device=d)

        def one():
            assert lib.cotr_op_expand(G.P(x), M1, G.P(wd), G.P(sd), G.P(bd), 0, G.P(yd), 256, G.P(w1), G.P(s1), G.P(b1), 1, G.P(y1), 64, s) == 0

        def two():
            assert lib.cotr_op_conv(G.P(x), G.P(wd), G.P(sd), G.P(bd), None, 0, G.P(yd2), B, 64, 64, 64, 256, 1, 1, s) == 0
            assert lib.cotr_op_conv(G.P(x), G.P(w1), G.P(s1), G.P(b1), None, 1, G.P(y12), B, 64, 64, 64, 64, 1, 1, s) == 0

        ta, tb = timeit(one), timeit(two)
        e1 = max(G.rel_err(yd, yd2), G.rel_err(y1, y12))
        by1 = M1 * (64 + 256 + 64) * 4
        print(f'{B:4d} | {ta:7.1f} ({by1 / ta * 1e-6:4.2f}) | {tb:7.1f} | {e1:.1e}')


if __name__ == '__main__':
    main()
