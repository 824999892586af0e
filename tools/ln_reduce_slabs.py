"""How much of an ln_reduce launch is its partial slabs?  (round-3 verdict weak 5: "two heads per attention workgroup / two chunks per
FFN workgroup was argued away, not measured".)  The launch that sums np partial outputs [np][rows][256] + bias + residual + LayerNorm,
timed in a captured chain for np = 2 ... 16 at the forward's row counts: the difference between np = 8 and np = 4 (16 and 8) is ALL
that halving the slabs could give that launch - the producing kernels would pay for it with twice the weights per workgroup.
GPU box:  python tools/ln_reduce_slabs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: t.data_ptr()
CHAIN = 64
import itertools
for rows, nbuf in itertools.product((1000, 512), (1, 8)):          # nbuf 8: the slabs rotate through 8 buffers = L2-cold, as in the forward
    out = []
    for np_ in (1, 2, 4, 8, 16):
        bufs = [torch.randn(np_, rows, 256, device=dev) for _ in range(nbuf)]
        parts = bufs[0]
        bias, w, b = torch.randn(256, device=dev), torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
        res = torch.randn(rows, 256, device=dev)
        ys = [torch.empty(rows, 256, device=dev) for _ in range(2)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            sp = s.cuda_stream
            for i in range(3):
                lib.cotr_op_ln_reduce(P(parts), np_, P(bias), P(res), P(w), P(b), P(ys[i & 1]), rows, sp)
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(CHAIN):
                    lib.cotr_op_ln_reduce(P(bufs[i % nbuf]), np_, P(bias), P(ys[(i + 1) & 1]) if i else P(res), P(w), P(b), P(ys[i & 1]), rows, sp)
            g.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(5):
                e0.record(s)
                for _ in range(4):
                    g.replay()
                e1.record(s); s.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1000 / (4 * CHAIN))
        out.append((np_, best))
    print(f'ln_reduce, {rows} rows, dependent chain, slabs {"L2-hot" if nbuf == 1 else "L2-cold (8 buffers in rotation)"}: ' + '  '.join(f'np={n}: {t:5.2f} us ({n * rows * 1024 / 1e6:4.1f} MB)' for n, t in out), flush=True)
