"""A/B of the attention sub-layer for many rows: ONE launch (att_rows.hip) against [q projection +] attention + out projection
(+ residual) + LayerNorm (the launches it replaces, on their tuned configurations).  Microseconds per sub-layer.

    python tools/bench_att_rows.py            -> profiles/r5_ab_att_rows.txt is its output on the MI355X
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotr_amd import _lib  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402
from tools.bench_ffn_rows import timeit  # noqa: E402


def main():
    lib = _lib.load_library()
    d = G.dev()
    g = torch.Generator().manual_seed(0)
    scale = 32 ** -0.5
    wq, bq = (torch.randn(256, 256, generator=g) / 16).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
    wo, bo = (torch.randn(256, 256, generator=g) / 16).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
    lw, lb = (torch.rand(256, generator=g) + 0.5).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
    print('# pairs x queries, mode | one launch us | separate launches us | max rel diff')
    for nb, nq, qp in ((32, 512, False), (64, 512, False), (32, 1000, True), (1, 32768, True), (16, 512, False), (8, 1000, True)):
        R = nb * nq
        kv = torch.randn(nb * 512, 3072, generator=g).to(d)
        kp, vp = ctypes.c_void_p(kv.data_ptr()), ctypes.c_void_p(kv.data_ptr() + 256 * 4)
        res = torch.randn(R, 256, generator=g).to(d)
        x2 = torch.randn(R, 256, generator=g).to(d)
        qkv = (torch.randn(R, 768, generator=g) * 0.5).to(d)
        y, y3 = torch.empty(R, 256, device=d), torch.empty(R, 256, device=d)
        qb, ao, tmp = torch.empty(R, 256, device=d), torch.empty(R, 256, device=d), torch.empty(R, 256, device=d)
        s = G.sptr()

        def one():
            if qp:
                rc = lib.cotr_op_att_rows(None, 0, G.P(res), G.P(x2), G.P(wq), G.P(bq), scale, kp, vp, 3072, G.P(wo), G.P(bo), G.P(res), G.P(lw),
                                          G.P(lb), G.P(y), nb, nq, s)
            else:
                rc = lib.cotr_op_att_rows(G.P(qkv), 768, None, None, None, None, 0.0, kp, vp, 3072, G.P(wo), G.P(bo), G.P(res), G.P(lw), G.P(lb),
                                          G.P(y), nb, nq, s)
            assert rc == 0, rc

        def sep():
            if qp:   # q = Wq(x + x2) * scale: the x + x2 prologue of the register-staged kernels (the decoder's q projection)
                # (the op-level entry has no q scale: same launch, same time - the comparison of values is left to the op test)
                assert lib.cotr_op_linear(G.P(res), G.P(x2), 0, G.P(wq), None, G.P(bq), None, 0, G.P(qb), R, 256, 256, s) == 0
                assert lib.cotr_op_attention(G.P(qb), 256, kp, vp, 3072, G.P(ao), 256, nb, nq, s) == 0
            else:
                assert lib.cotr_op_attention(G.P(qkv), 768, kp, vp, 3072, G.P(ao), 256, nb, nq, s) == 0
            assert lib.cotr_op_linear(G.P(ao), None, 0, G.P(wo), None, G.P(bo), G.P(res), 0, G.P(tmp), R, 256, 256, s) == 0
            assert lib.cotr_op_layernorm(G.P(tmp), G.P(lw), G.P(lb), G.P(y3), R, s) == 0

        t1, t3 = timeit(one), timeit(sep)
        diff = float('nan') if qp else G.rel_err(y, y3)
        fl = R * (2 * 2 * 512 * 256 + 2 * 256 * 256 * (2 if qp else 1))
        print(f'{nb:3d} x {nq:5d} {"dec (q proj)" if qp else "enc (q given)"} | {t1:8.1f} ({fl / t1 * 1e-6:6.1f} TFLOP/s, {fl / t1 * 1e-6 / 157.3:.3f}) | '
              f'{t3:8.1f} ({fl / t3 * 1e-6:6.1f}, {fl / t3 * 1e-6 / 157.3:.3f}) | {diff:.2e}')


if __name__ == '__main__':
    main()
