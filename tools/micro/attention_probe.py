"""Attention kernel time per key-split setting at the batched shapes. GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: ctypes.c_void_p(t.data_ptr())

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for tag, nb, nq in (('encoder, 32 pairs', 32, 512), ('decoder chunk, dense pass', 1, 32768), ('decoder, 32 pairs x 1000', 32, 1000),
                    ('encoder, 1 pair', 1, 512), ('decoder, 1 pair x 1000', 1, 1000), ('encoder, 4 pairs', 4, 512)):
    q = torch.randn(nb * nq, 256, device='cuda') * 0.3
    k = torch.randn(nb * 512, 256, device='cuda'); v = torch.randn(nb * 512, 256, device='cuda')
    o = torch.empty(nb * nq, 256, device='cuda')
    ref = None
    row = []
    for ns in (1, 2, 4, 8, 16):
        _lib.set_knob('attention_splits', ns)
        sp = _lib.current_stream_ptr()
        call = lambda: lib.cotr_op_attention(P(q), 256, P(k), P(v), 256, P(o), 256, nb, nq, sp)
        assert call() == 0
        torch.cuda.synchronize()
        if ref is None: ref = o.clone()
        err = float((o - ref).abs().max())
        row.append((ns, timeit(call), err))
    _lib.set_knob('attention_splits', 0)
    flop = nb * nq * 512 * 32 * 2 * 2 * 8
    print(f'{tag:28s} nb={nb:3d} nq={nq:6d}: ' + '  '.join(f'ns{ns}: {t:7.1f}us' for ns, t, _ in row) +
          f'   best {flop / min(t for _, t, _ in row) / 1e6:6.1f} TFLOP/s  max|diff vs ns1| {max(e for _, _, e in row):.1e}', flush=True)
