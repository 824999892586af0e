"""attention over many rows: the 64-query kernel (attention_wide_kernel) against K_h / V_h resident in LDS (attention_res_kernel), per
shape, HIP events over back-to-back launches.  GPU box.   python tools/ab_attention_resident.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: t.data_ptr()
for name, nb, nq in (('encoder, 32 pairs x 512', 32, 512), ('decoder, 32 pairs x 1000', 32, 1000), ('encoder, 64 x 512', 64, 512),
                     ('decoder, 64 x 1000', 64, 1000), ('dense pass, 1 x 32768 (one decode chunk)', 1, 32768), ('8 x 1000', 8, 1000), ('4 x 1000', 4, 1000), ('16 x 512', 16, 512), ('2 x 4096', 2, 4096)):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(nb * nq, 256, generator=g).cuda()
    kv = torch.randn(nb * 512, 512, generator=g).cuda()
    o = torch.empty(nb * nq, 256, device='cuda')
    res = []
    for resident in (0, 1):
        _lib.set_knob('attention_resident', resident)
        run = lambda: lib.cotr_op_attention(P(q), 256, P(kv), P(kv[:, 256:]), 512, P(o), 256, nb, nq, _lib.current_stream_ptr())
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    gf = 4.0 * nb * nq * 512 * 256 / 1e9
    print(f'{name:42s} 64-query kernel {res[0]:7.1f} us ({gf / res[0] * 1e3:5.1f} TFLOP/s)   resident K/V {res[1]:7.1f} us ({gf / res[1] * 1e3:5.1f} TFLOP/s)', flush=True)
