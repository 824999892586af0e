"""RESEARCH (round-3 verdict item 8): the large-tile GEMM on packed split-f16 operands (configurations 46 / 47 of libcotr_hip_exp.so,
csrc/experimental/gemm_h2.h: three v_mfma_f32_32x32x16_f16 per fp32 product) next to the fp32-MFMA configurations 26 / 27 on the
shapes of the batched forward and on plain squares; the cost of packing an operand (cotr_op_split_h2) is timed on its own.
"TFLOP/s" counts the fp32 work (2 M N K) whatever the kernel does inside.   GPU box:  COTR_HIP_EXPERIMENTAL=1 python tools/bench_split_f16.py"""
import ctypes, os, sys
os.environ.setdefault('COTR_HIP_EXPERIMENTAL', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: t.data_ptr()
shapes = [(16384, 3072, 256), (32000, 256, 256), (16384, 256, 256), (4096, 4096, 4096), (8192, 8192, 8192), (16384, 1024, 256), (16384, 256, 1024), (32000, 1024, 256), (32000, 256, 1024),
          (16384, 256, 2304), (65536, 128, 1152), (262144, 64, 576), (262144, 256, 64), (262144, 64, 256), (16384, 768, 256)]
for M, N, K in shapes:
    x = torch.relu(torch.randn(M, K, device=dev))
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(M, N, device=dev)
    xp, wp = torch.empty_like(x), torch.empty_like(w)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.cotr_op_split_h2(P(x), P(xp), x.numel(), s) == 0 and lib.cotr_op_split_h2(P(w), P(wp), w.numel(), s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.cotr_op_split_h2(P(x), P(xp), x.numel(), s)
    e1.record()
    torch.cuda.synchronize()
    pack_us = e0.elapsed_time(e1) * 100
    line = f'{M:7d} x {N:5d} x {K:5d}:'
    t = {}
    for cfg, a_, w_ in ((26, x, w), (27, x, w), (40, x, w), (46, xp, wp), (47, xp, wp), (48, xp, wp), (49, xp, wp), (50, xp, wp), (51, xp, wp)):
        us = ctypes.c_float(0)
        r = lib.cotr_bench_linear(P(a_), P(w_), None, P(y), M, N, K, cfg, 10, ctypes.byref(us))
        t[cfg] = us.value if r == 0 else float('nan')
        line += f'  {cfg}: {t[cfg]:7.1f} us {2.0 * M * N * K / t[cfg] / 1e6:5.0f} TF'
    nn = lambda v: v if v == v else 1e30
    best32, best16 = min(nn(t[26]), nn(t[27]), nn(t[40])), min(nn(t[c]) for c in (46, 47, 48, 49, 50, 51))
    print(line + f'   packing x: {pack_us:6.1f} us   split-f16 / fp32 = {best32 / best16:.2f}x ({best32 / (best16 + pack_us):.2f}x with the packing pass)', flush=True)

# the convolutions of the batched backbone (32 pairs), implicit GEMM on packed pixels / weights
print('convolutions (B pairs, H = W of a half, Cin -> Cout, k, stride): us per launch')
for B, H, cin, cout, k, stride in ((32, 64, 64, 64, 3, 1), (32, 32, 128, 128, 3, 1), (32, 16, 256, 256, 3, 1), (32, 64, 256, 128, 1, 1), (32, 64, 256, 512, 1, 2),
                                   (32, 32, 512, 256, 1, 1), (32, 64, 64, 256, 1, 1), (32, 16, 1024, 256, 1, 1), (32, 16, 256, 1024, 1, 1)):
    x = torch.relu(torch.randn(B, H, 2 * H, cin, device=dev))
    w = torch.randn(cout, k * k * cin, device=dev) / (k * k * cin) ** 0.5
    sc, bi = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    Ho = H // stride
    y = torch.empty(B, Ho, 2 * Ho, cout, device=dev)
    xp, wp = torch.empty_like(x), torch.empty_like(w)
    s = torch.cuda.current_stream().cuda_stream
    lib.cotr_op_split_h2(P(x), P(xp), x.numel(), s); lib.cotr_op_split_h2(P(w), P(wp), w.numel(), s)
    line = f'{B} x {H}x{2 * H} {cin:4d} -> {cout:4d} k{k} /{stride}:'
    for cfg, a_, w_ in ((26, x, w), (27, x, w), (40, x, w), (46, xp, wp), (47, xp, wp), (48, xp, wp), (49, xp, wp), (51, xp, wp)):
        us = ctypes.c_float(0)
        r = lib.cotr_bench_conv(P(a_), P(w_), P(sc), P(bi), P(y), B, H, H, cin, cout, k, stride, cfg, 10, ctypes.byref(us))
        line += f'  {cfg}: {us.value if r == 0 else float("nan"):7.1f}'
    print(line, flush=True)
