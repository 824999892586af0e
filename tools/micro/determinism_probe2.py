import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cotr_amd import _lib
lib = _lib.load_library()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
g = torch.Generator().manual_seed(0)
for M, N, K in ((65536, 512, 128), (262144, 64, 256)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
    ref = torch.empty(M, N, device='cuda')
    lib.cotr_op_linear_cfg(P(x), P(w), P(b), None, 0, P(ref), M, N, K, 2, _lib.current_stream_ptr())
    torch.cuda.synchronize()
    for it in range(4):
        y = torch.full((M, N), float('nan'), device='cuda')
        lib.cotr_op_linear_cfg(P(x), P(w), P(b), None, 0, P(y), M, N, K, 27, _lib.current_stream_ptr())
        torch.cuda.synchronize()
        d = (y - ref).abs()
        badmask = d > 1e-3
        rows = badmask.any(dim=1).nonzero().flatten()
        cols = badmask.any(dim=0).nonzero().flatten()
        exact = int((y != ref).sum())
        print(f'M={M} N={N} K={K} run{it}: max|diff vs cfg2|={float(d.max()):.3e} elements != cfg2: {exact}  >1e-3: {int(badmask.sum())} '
              f'rows {rows[:8].tolist()}.. ({len(rows)}) cols {cols[:8].tolist()}.. ({len(cols)})', flush=True)
