"""The 1x1 expansion of every bottleneck (conv3: C -> 4C, + identity, ReLU) carries a residual that tools/tune_gemm.py
does not time; with many pairs per pass these launches are bound by exactly that traffic.  Time them WITH the residual
(stream launches between torch events; the kernels are long enough for launch overhead not to matter) and print
table lines for cotr_amd/csrc/gemm_tuned.inc.   python tools/tune_expansions.py [pairs,...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
pairs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,8,16,32').split(',')]
CFGS = [0, 1, 2, 26, 27]
g = torch.Generator().manual_seed(0)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for B in pairs:
    for hin, cin in ((64, 64), (32, 128), (16, 256)):
        cout = 4 * cin
        x = torch.randn(B, hin, 2 * hin, cin, generator=g).cuda()
        w = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).cuda()
        sc, bi = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
        r = torch.randn(B, hin, 2 * hin, cout, generator=g).cuda()
        y = torch.empty(B, hin, 2 * hin, cout, device='cuda')
        M = B * hin * 2 * hin
        row = {}
        for c in CFGS:
            sp = _lib.current_stream_ptr()
            call = lambda: lib.cotr_op_conv_cfg(P(x), P(w), P(sc), P(bi), P(r), 1, P(y), B, hin, hin, cin, cout, 1, 1, c, sp)
            if call() != 0:
                continue
            row[c] = timeit(call)
        best = min(row, key=row.get)
        reg = min((c for c in row if c < 26), key=row.get)
        print(f'{{1, {M}, {cout}, {cin}, {best}, {reg}}},  // {row[best]:.2f} us  (with residual: ' +
              ', '.join(f'cfg{c} {t:.1f}' for c, t in row.items()) + ')', flush=True)
