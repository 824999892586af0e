"""Fixed cost per tile vs cost per K step of the large-tile GEMMs: time M x N x K at fixed (M, N) for K = 128 ... 8192 and fit
t = rounds * (T0 + KT * Ts).  If T0 is several K steps, a kernel that pipelines ACROSS tiles (persistent, loaders running ahead
into the next tile, the epilogue of tile i beside the MFMAs of tile i+1) has that much to win on the K <= 1024 shapes of the
forward; if it is small, the loss is elsewhere.   GPU box:  python tools/tile_overhead.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cotr_amd import _lib
lib = _lib.load_library()
dev = torch.device('cuda:0')
P = lambda t: t.data_ptr()
PP = (44, 45) if _lib.experimental_selected() else ()      # experimental: 26 / 27 with the LDS-free epilogue (42 / 43: gemm_pp.hip)
SHAPES = (('16384 x 1024', 16384, 1024, (26, 27, 40) + PP), ('16384 x 256', 16384, 256, (26, 27, 40, 41) + PP), ('32000 x 256', 32000, 256, (26, 27) + PP),
          ('65536 x 512', 65536, 512, (26, 27)), ('262144 x 256', 262144, 256, (26, 27) + PP))
if len(sys.argv) > 2:
    SHAPES = (('16384 x 1024', 16384, 1024, (26, 27) + PP), ('262144 x 256', 262144, 256, (26, 27) + PP))
if len(sys.argv) > 1:                     # knob ws_flags (process-wide set): bit 2 = the persistent kernel skips its stores (timing experiment)
    _lib.set_knob('ws_flags', int(sys.argv[1]))
    print('ws_flags =', sys.argv[1])
for tag, M, N, cfgs in SHAPES:
    for cfg in cfgs:
        ks, ts = [], []
        for K in (64, 256, 1024, 2048, 4096):
            if M * K * 4 > (6 << 30):
                continue
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) / K ** 0.5
            y = torch.empty(M, N, device=dev)
            us = ctypes.c_float(0)
            if lib.cotr_bench_linear(P(x), P(w), None, P(y), M, N, K, cfg, 10, ctypes.byref(us)) != 0:
                continue
            ks.append(K); ts.append(us.value)
            del x, w, y
        if len(ks) < 3:
            continue
        bn = 128 if cfg in (26, 40, 44) else 64
        tiles = (M // 128) * (N // bn)
        per_cu = tiles / 256.0
        kt = np.array(ks) / 32.0
        A = np.stack([np.ones_like(kt), kt], 1)
        (t0, tstep), *_ = np.linalg.lstsq(A[-4:], np.array(ts)[-4:], rcond=None)       # fit on the K-deep end
        mfma_step = 128 * bn * 32 * 2 / (157.3e12 / 256) * 1e6 * per_cu                # us per K step of all of a CU's tiles at the peak
        print(f'{tag} cfg {cfg} ({tiles} tiles of 128x{bn}, {per_cu:.1f} per CU): ' + '  '.join(f'K={k}: {t:7.1f}us {2.0 * M * N * k / t / 1e6:5.1f}TF' for k, t in zip(ks, ts)))
        print(f'      fit t = {t0:6.1f} us + {tstep:6.3f} us per K step (the MFMA peak allows {mfma_step:6.3f}: {mfma_step / tstep:.3f} in steady state); '
              f'fixed part = {t0 / tstep:4.1f} K steps = {t0 / per_cu:5.2f} us per tile and CU', flush=True)
