"""Large-tile GEMM configurations (26, 27: gemm_big.hip) against the 64x64 / 128x128 register-staged ones on the
contraction shapes of the batched regime (32 pairs per backbone pass).  GPU box.
    python tools/bench_big.py [pairs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib

lib = _lib.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CFGS = [2, 26, 27, 28, 29, 40, 41]
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
us = ctypes.c_float()
g = torch.Generator().manual_seed(0)
dev = torch.device('cuda:0')
convs = [(64, 64, 256, 1, 1), (64, 256, 64, 1, 1), (64, 64, 64, 3, 1), (64, 256, 128, 1, 1), (64, 128, 128, 3, 2),
         (32, 128, 512, 1, 1), (32, 512, 128, 1, 1), (32, 128, 128, 3, 1), (64, 256, 512, 1, 2), (32, 512, 256, 1, 1),
         (32, 256, 256, 3, 2), (16, 256, 1024, 1, 1), (16, 1024, 256, 1, 1), (16, 256, 256, 3, 1), (32, 512, 1024, 1, 2)]
print('conv (hin, cin, cout, k, stride)   M N K   ' + '  '.join(f'cfg{c:>2}' for c in CFGS) + '   best TFLOP/s')
for hin, cin, cout, k, stride in convs:
    pad = k // 2
    ho = (hin + 2 * pad - k) // stride + 1
    x = torch.randn(B, hin, 2 * hin, cin, generator=g).to(dev)
    w = (torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5).to(dev)
    sc, bi = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    y = torch.empty(B, ho, 2 * ho, cout, device=dev)
    M, N, K = B * ho * 2 * ho, cout, k * k * cin
    lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, hin, hin, cin, cout, k, stride, 2, None)
    torch.cuda.synchronize()
    ref = y.clone()
    row = []
    for c in CFGS:
        y.fill_(float('nan'))
        rc = lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), None, 1, P(y), B, hin, hin, cin, cout, k, stride, c, None)
        torch.cuda.synchronize()
        if rc != 0:
            row.append(None)
            continue
        err = float((y - ref).abs().max() / ref.abs().max())
        if not err < 1e-4:
            row.append(-1.0)
            continue
        rc = lib.cotr_bench_conv(P(x), P(w), P(sc), P(bi), P(y), B, hin, hin, cin, cout, k, stride, c, 20, ctypes.byref(us))
        row.append(us.value if rc == 0 else None)
    ok = [t for t in row if t and t > 0]
    print(f'{(hin, cin, cout, k, stride)!s:28s} {M:7d} {N:5d} {K:5d}  ' +
          '  '.join('   n/a' if t is None else (' WRONG' if t < 0 else f'{t:6.1f}') for t in row) +
          f'   {2.0 * M * N * K / min(ok) / 1e6:6.1f}', flush=True)
lins = [(512 * B, 256, 1024), (512 * B, 768, 256), (512 * B, 256, 256), (512 * B, 1024, 256), (512 * B, 3072, 256),
        (32768, 256, 256), (32768, 1024, 256), (32768, 256, 1024)]
print('linear M N K')
for M, N, K in lins:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bi = torch.randn(N, generator=g).to(dev)
    y = torch.empty(M, N, device=dev)
    lib.cotr_op_linear_cfg(P(x), P(w), P(bi), None, 0, P(y), M, N, K, 2, None)
    torch.cuda.synchronize()
    ref = y.clone()
    row = []
    for c in CFGS:
        y.fill_(float('nan'))
        rc = lib.cotr_op_linear_cfg(P(x), P(w), P(bi), None, 0, P(y), M, N, K, c, None)
        torch.cuda.synchronize()
        if rc != 0:
            row.append(None)
            continue
        err = float((y - ref).abs().max() / ref.abs().max())
        if not err < 1e-4:
            row.append(-1.0)
            continue
        rc = lib.cotr_bench_linear(P(x), P(w), P(bi), P(y), M, N, K, c, 20, ctypes.byref(us))
        row.append(us.value if rc == 0 else None)
    ok = [t for t in row if t and t > 0]
    print(f'{"":28s} {M:7d} {N:5d} {K:5d}  ' +
          '  '.join('   n/a' if t is None else (' WRONG' if t < 0 else f'{t:6.1f}') for t in row) +
          f'   {2.0 * M * N * K / min(ok) / 1e6:6.1f}', flush=True)
