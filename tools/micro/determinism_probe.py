"""Bitwise repeatability of the large-tile GEMM configurations. GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(0)
bad = 0
for M, N, K in ((262144, 256, 64), (65536, 512, 128), (16384, 1024, 256), (262144, 64, 256), (16384, 256, 2304), (4133, 256, 256)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda()
    for res in (None, r):
        for c in (2, 26, 27, 28, 29):
            outs = []
            for it in range(6):
                y = torch.full((M, N), float('nan'), device='cuda')
                rc = lib.cotr_op_linear_cfg(P(x), P(w), P(b), P(res), 1, P(y), M, N, K, c, _lib.current_stream_ptr())
                if rc != 0:
                    break
                torch.cuda.synchronize()
                outs.append(y)
            if not outs:
                continue
            diff = [int((o != outs[0]).sum()) for o in outs[1:]]
            nan = int(torch.isnan(outs[0]).sum())
            if any(diff) or nan:
                bad += 1
            print(f'linear M={M} N={N} K={K} res={int(res is not None)} cfg{c}: mismatching elements vs run 0: {diff} nan={nan}', flush=True)
for B, hin, cin, cout, k, stride in ((32, 64, 64, 64, 3, 1), (32, 64, 128, 128, 3, 2), (8, 32, 128, 128, 3, 1), (32, 64, 256, 512, 1, 2)):
    x = torch.randn(B, hin, 2 * hin, cin, generator=g).cuda()
    w = (torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5).cuda()
    sc, bi = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    ho = (hin + 2 * (k // 2) - k) // stride + 1
    for c in (2, 26, 27, 28, 29):
        outs = []
        for it in range(6):
            y = torch.full((B, ho, 2 * ho, cout), float('nan'), device='cuda')
            rc = lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, hin, hin, cin, cout, k, stride, c, _lib.current_stream_ptr())
            if rc != 0:
                break
            torch.cuda.synchronize()
            outs.append(y)
        if not outs:
            continue
        diff = [int((o != outs[0]).sum()) for o in outs[1:]]
        nan = int(torch.isnan(outs[0]).sum())
        if any(diff) or nan:
            bad += 1
        print(f'conv {(B, hin, cin, cout, k, stride)} cfg{c}: mismatching elements vs run 0: {diff} nan={nan}', flush=True)
print('BAD' if bad else 'ALL REPEATABLE', bad)
