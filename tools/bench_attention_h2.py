"""RESEARCH: the split-f16 attention kernel (csrc/experimental/attention_h2.hip) next to the fp32 kernels at the batched shapes.
GPU box:  COTR_HIP_EXPERIMENTAL=1 python tools/bench_attention_h2.py"""
import os, sys
os.environ.setdefault('COTR_HIP_EXPERIMENTAL', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: t.data_ptr()
for nb, nq in ((32, 512), (32, 1000), (64, 512), (1, 131072 // 4)):
    q = torch.randn(nb * nq, 256, device=dev) / 32 ** 0.5
    kv = torch.randn(nb * 512, 512, device=dev)
    qp, kvp = torch.empty_like(q), torch.empty_like(kv)
    s = torch.cuda.current_stream().cuda_stream
    lib.cotr_op_split_h2(P(q), P(qp), q.numel(), s); lib.cotr_op_split_h2(P(kv), P(kvp), kv.numel(), s)
    o = torch.empty(nb * nq, 256, device=dev)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t32 = t(lambda: lib.cotr_op_attention(P(q), 256, P(kv), P(kv[:, 256:]), 512, P(o), 256, nb, nq, s))
    t16 = t(lambda: lib.cotr_op_attention_h2(P(qp), 256, 1, P(kvp), P(kvp[:, 256:]), 512, P(o), 256, 1, nb, nq, s))
    t16f = t(lambda: lib.cotr_op_attention_h2(P(q), 256, 0, P(kvp), P(kvp[:, 256:]), 512, P(o), 256, 0, nb, nq, s))
    fl = 4.0 * nb * nq * 512 * 256
    print(f'{nb} pairs x {nq} queries: fp32 kernel {t32:7.1f} us ({fl / t32 / 1e6:5.0f} TF)   split-f16 packed q/o {t16:7.1f} us ({fl / t16 / 1e6:5.0f} TF)   fp32 q/o {t16f:7.1f} us   {t32 / t16:.2f}x', flush=True)
