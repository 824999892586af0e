"""Phase timestamps of the fused FFN kernel (cotr_debug_ffn_times) at the one-pair row counts.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
names = ['entry', 'loads issued', 'X+W1_0 usable', 'H_0 done', 'phase2_0 issued', 'W1_1 usable', 'loop done', 'stored']
for M in (512, 1000):
    x = torch.randn(M, 256, device='cuda')
    w1, b1 = torch.randn(1024, 256, device='cuda') / 16, torch.randn(1024, device='cuda')
    w2, b2 = torch.randn(256, 1024, device='cuda') / 32, torch.randn(256, device='cuda')
    lw, lb = torch.ones(256, device='cuda'), torch.zeros(256, device='cuda')
    nch = lib.cotr_op_ffn_chunks(M)
    scratch = torch.empty(nch * M * 256, device='cuda')
    y = torch.empty(M, 256, device='cuda')
    times = torch.zeros(4096, 8, dtype=torch.int64, device='cuda')
    run = lambda: lib.cotr_op_ffn_block(P(x), P(w1), P(b1), P(w2), P(b2), P(lw), P(lb), P(scratch), P(y), M, sp)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    lib.cotr_debug_ffn_times(P(times))
    times.zero_()
    torch.cuda.synchronize()
    assert run() == 0
    torch.cuda.synchronize()
    lib.cotr_debug_ffn_times(None)
    t = times.cpu()
    t = t[t[:, 0] > 0].double()
    t0 = t[:, 0].min()
    line = '  '.join(f'{n} {((t[:, i] - t0) * 0.01)[t[:, i] > 0].mean():5.2f}' for i, n in enumerate(names) if (t[:, i] > 0).any())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f'M={M} x{nch} chunks, {t.shape[0]} wgs, FFN + ln_reduce {e0.elapsed_time(e1) * 20:.2f} us back-to-back | us since first entry: {line}', flush=True)
