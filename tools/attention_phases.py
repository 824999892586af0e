"""Phase timestamps of the fused attention kernel (cotr_debug_attention_times) at the one-pair shapes.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
names = ['entry', 'q ready', 'key loop done', 'merged', 'out-proj staged', 'stored']
for name, nq, qp in (('encoder (q given, out-proj epilogue)', 512, False), ('decoder (q-proj prologue x + x2, out-proj epilogue)', 1000, True)):
    x, x2 = torch.randn(nq, 256, device='cuda'), torch.randn(nq, 256, device='cuda')
    wq, bq = torch.randn(256, 256, device='cuda') / 16, torch.randn(256, device='cuda')
    wo = torch.randn(256, 256, device='cuda') / 16
    kv = torch.randn(512, 512, device='cuda')
    q = torch.randn(nq, 256, device='cuda')
    part = torch.empty(8, nq, 256, device='cuda')
    times = torch.zeros(4096, 8, dtype=torch.int64, device='cuda')
    if qp:
        run = lambda: lib.cotr_op_attention_fused(None, 0, P(x), P(x2), P(wq), P(bq), 32 ** -0.5, P(kv), P(kv[:, 256:]), 512, None, 0, P(wo), P(part), 1, nq, sp)
    else:
        run = lambda: lib.cotr_op_attention_fused(P(q), 256, None, None, None, None, 0.0, P(kv), P(kv[:, 256:]), 512, None, 0, P(wo), P(part), 1, nq, sp)
    for _ in range(5):
        assert run() == 0
    torch.cuda.synchronize()
    lib.cotr_debug_attention_times(P(times)); times.zero_(); torch.cuda.synchronize()
    run(); torch.cuda.synchronize()
    lib.cotr_debug_attention_times(None)
    t = times.cpu(); t = t[t[:, 0] > 0].double(); t0 = t[:, 0].min()
    line = '  '.join(f'{n} {((t[:, i] - t0) * 0.01)[t[:, i] > 0].mean():5.2f}' for i, n in enumerate(names) if (t[:, i] > 0).any())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {t.shape[0]} wgs, {e0.elapsed_time(e1) * 20:.2f} us back-to-back | us since first entry: {line}', flush=True)
