"""The plain Linear GEMMs of the one-pair forward under every launch configuration (back-to-back launch time) and the pick."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else t.data_ptr()
sp = _lib.current_stream_ptr()
us = ctypes.c_float()
for name, M, N, K, relu in [('input_proj', 512, 256, 1024, 0), ('corr_embed 0/1', 1000, 256, 256, 1), ('enc QKV (no pos)', 512, 768, 256, 0),
                            ('dec K/V (no pos)', 512, 3072, 256, 0)]:
    x, w, b = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') / K ** 0.5, torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    out = []
    for cfg in range(lib.cotr_gemm_num_configs()):
        if lib.cotr_op_linear_cfg(P(x), P(w), P(b), None, relu, P(y), M, N, K, cfg, sp) != 0:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            lib.cotr_op_linear_cfg(P(x), P(w), P(b), None, relu, P(y), M, N, K, cfg, sp)
        e1.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(e1) * 10, cfg))
    lib.cotr_bench_linear(P(x), P(w), P(b), P(y), M, N, K, -1, 50, ctypes.byref(us))
    out.sort()
    print(f'{name:18s} {M}x{N}x{K}: library pick {us.value:.2f} us (graph-paced) | ' + '  '.join(f'cfg{c} {u:.2f}' for u, c in out[:6]), flush=True)
