"""A/B of layer1's conv2 -> conv3 for many pairs: ONE launch (conv23.hip) against the two tuned launches it replaces (3x3 conv + FrozenBN +
ReLU, 1x1 expansion + FrozenBN + identity + ReLU).  Prints microseconds per block and TFLOP/s of the fp32-MFMA peak (157.3).

    python tools/bench_conv23.py [pairs ...]          -> profiles/r5_ab_conv23.txt is its output on the MI355X
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
    w2 = (torch.randn(64, 3, 3, 64, generator=g) / math.sqrt(576)).to(d)
    w3 = (torch.randn(256, 64, generator=g) / 8).to(d)
    s2, b2 = (torch.rand(64, generator=g) + 0.5).to(d), torch.randn(64, generator=g).to(d)
    s3, b3 = (torch.rand(256, generator=g) + 0.5).to(d), torch.randn(256, generator=g).to(d)
    w2m = (torch.randn(128, 3, 3, 128, generator=g) / math.sqrt(1152)).to(d)
    w3m = (torch.randn(512, 128, generator=g) / math.sqrt(128)).to(d)
    s2m, b2m = (torch.rand(128, generator=g) + 0.5).to(d), torch.randn(128, generator=g).to(d)
    s3m, b3m = (torch.rand(512, generator=g) + 0.5).to(d), torch.randn(512, generator=g).to(d)
    print('# pairs | layer1: one launch us (TFLOP/s, of peak) | two launches us (TFLOP/s, of peak) | max rel diff || layer2: one launch us (of peak) | two launches us (of peak) | max rel diff')
    for B in pairs:
        t1 = torch.relu(torch.randn(B, 64, 128, 64, generator=g)).to(d)
        idt = torch.randn(B, 64, 128, 256, generator=g).to(d)
        y = torch.empty(B, 64, 128, 256, device=d)
        t2 = torch.empty(B, 64, 128, 64, device=d)
        y2 = torch.empty(B, 64, 128, 256, device=d)
        s = G.sptr()

        def one():
            assert lib.cotr_op_conv23(G.P(t1), G.P(w2), G.P(s2), G.P(b2), G.P(w3), G.P(s3), G.P(b3), G.P(idt), G.P(y), B, s) == 0

        def two():
            assert lib.cotr_op_conv(G.P(t1), G.P(w2), G.P(s2), G.P(b2), None, 1, G.P(t2), B, 64, 64, 64, 64, 3, 1, s) == 0
            assert lib.cotr_op_conv(G.P(t2), G.P(w3), G.P(s3), G.P(b3), G.P(idt), 1, G.P(y2), B, 64, 64, 64, 256, 1, 1, s) == 0

        ta, tb = timeit(one), timeit(two)
        fl = 2 * B * 8192 * 64 * (576 + 256)
        line = (f'{B:4d} | {ta:8.1f} ({fl / ta * 1e-6:6.1f}, {fl / ta * 1e-6 / 157.3:.3f}) | {tb:8.1f} ({fl / tb * 1e-6:6.1f}, {fl / tb * 1e-6 / 157.3:.3f}) | '
                f'{G.rel_err(y, y2):.2e}')
        # layer2 (conv23m.hip): 128 -> 128 (3x3, stride 1) -> 512
        u1 = torch.relu(torch.randn(B, 32, 64, 128, generator=g)).to(d)
        idm = torch.randn(B, 32, 64, 512, generator=g).to(d)
        ym, u2, ym2 = torch.empty(B, 32, 64, 512, device=d), torch.empty(B, 32, 64, 128, device=d), torch.empty(B, 32, 64, 512, device=d)

        def onem():
            assert lib.cotr_op_conv23m(G.P(u1), G.P(w2m), G.P(s2m), G.P(b2m), G.P(w3m), G.P(s3m), G.P(b3m), G.P(idm), G.P(ym), B, 1, s) == 0

        def twom():
            assert lib.cotr_op_conv(G.P(u1), G.P(w2m), G.P(s2m), G.P(b2m), None, 1, G.P(u2), B, 32, 32, 128, 128, 3, 1, s) == 0
            assert lib.cotr_op_conv(G.P(u2), G.P(w3m), G.P(s3m), G.P(b3m), G.P(idm), 1, G.P(ym2), B, 32, 32, 128, 512, 1, 1, s) == 0

        tc, td = timeit(onem), timeit(twom)
        flm = 2 * B * 2048 * 128 * (1152 + 512)
        print(line + f' || {tc:8.1f} ({flm / tc * 1e-6 / 157.3:.3f}) | {td:8.1f} ({flm / td * 1e-6 / 157.3:.3f}) | {G.rel_err(ym, ym2):.2e}')


if __name__ == '__main__':
    main()
