"""fused FFN block (2 launches) vs linear1 + linear2 + layernorm (3 launches), GPU-paced via torch events over a chain."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
d = torch.device('cuda:0')
for M in (512, 1000, 2048, 4000):
    x = torch.randn(M, 256, device=d); w1 = torch.randn(1024, 256, device=d) / 16; b1 = torch.randn(1024, device=d)
    w2 = torch.randn(256, 1024, device=d) / 32; b2 = torch.randn(256, device=d); lw = torch.ones(256, device=d); lb = torch.zeros(256, device=d)
    scratch = torch.empty(lib.cotr_op_ffn_chunks(M) * M * 256, device=d); y = torch.empty(M, 256, device=d)
    hid = torch.empty(M, 1024, device=d); pre = torch.empty(M, 256, device=d)
    s = _lib.current_stream_ptr()
    def fused():
        lib.cotr_op_ffn_block(P(x), P(w1), P(b1), P(w2), P(b2), P(lw), P(lb), P(scratch), P(y), M, s)
    def unfused():
        lib.cotr_op_linear(P(x), None, 0, P(w1), None, P(b1), None, 1, P(hid), M, 1024, 256, s)
        lib.cotr_op_linear(P(hid), None, 0, P(w2), None, P(b2), P(x), 0, P(pre), M, 256, 1024, s)
        lib.cotr_op_layernorm(P(pre), P(lw), P(lb), P(y), M, s)
    for name, fn in (('fused', fused), ('unfused', unfused)):
        g = torch.cuda.CUDAGraph()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            s = _lib.current_stream_ptr()
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): fn()
            e1.record(); torch.cuda.synchronize()
        print(f'M={M} {name}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per block (chunks={lib.cotr_op_ffn_chunks(M)})', flush=True)
        s = _lib.current_stream_ptr()
