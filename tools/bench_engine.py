"""End-to-end zoom-in refinement throughput on the MI355X: ZoomEngine (one crop launch + batched model calls per
level) vs the reference's loop shape (32 tasks per call, PIL crops on the host, H2D per batch) with the same HIP model.
    python tools/bench_engine.py [n_queries] [--config2] [KNOB=INT ...]     (knobs of the model's handle, e.g. split_f16=3 with
    COTR_HIP_EXPERIMENTAL=1: the research path of docs/LABNOTES.md 3e at engine level)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cotr_amd
from cotr_amd.inference import ZoomEngine
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
from tests.engine_fixtures import synthetic_pair, pil_cropper_factory

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000
img_a, img_b = synthetic_pair(3, (783, 1064), (1053, 689))     # the cathedral demo pair's sizes
rng = np.random.default_rng(0)
loc_from = np.stack([rng.uniform(5, img_a.shape[1] - 5, n), rng.uniform(5, img_a.shape[0] - 5, n)], 1)
loc_to = np.stack([rng.uniform(5, img_b.shape[1] - 5, n), rng.uniform(5, img_b.shape[0] - 5, n)], 1)
zooms = np.linspace(0.5, 0.0625, 4)
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
for kv in [a for a in sys.argv[1:] if '=' in a]:
    m.set_knob(kv.split('=')[0], int(kv.split('=')[1]))
    print('knob', kv)
for max_pairs, tag in ((256, 'ZoomEngine, device crops, 256 crops per model call'),
                       (1024, 'ZoomEngine, device crops, 1024 crops per model call')):
    eng = ZoomEngine(m, max_pairs=max_pairs)
    eng.refine(img_a, img_b, loc_from[:64], loc_to[:64], 1.0, 1.0, zooms, 1, force=True)   # warm-up
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = eng.refine(img_a, img_b, loc_from, loc_to, 1.0, 1.0, zooms, 1, force=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'{tag}: {n} queries x {len(zooms)} levels in {dt:.3f} s = {n / dt:.0f} correspondences/s '
          f'({res.crops} crops, {res.model_calls} levels)', flush=True)
# reference loop shape: 32 tasks per call, host PIL crops + H2D (SparseEngine.form_batch / infer_batch)
n_ref = min(n, 128)
eng = ZoomEngine(m, max_pairs=32, make_cropper=lambda a, b, d: (lambda boxes, out, f=pil_cropper_factory(a, b, d): f(boxes, out)))
t = time.perf_counter()
eng.refine(img_a, img_b, loc_from[:n_ref], loc_to[:n_ref], 1.0, 1.0, zooms, 1, force=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f'reference loop shape (32 per call, PIL crops on the host, same HIP model): {n_ref} queries in {dt:.3f} s = '
      f'{n_ref / dt:.0f} correspondences/s', flush=True)
# dense initial pass (cotr_flow): 4 patch pairs x 131072 queries in one model call + device post-processing
eng = ZoomEngine(m)
eng.flow(img_a[:300, :420], img_b[:350, :330])                   # warm-up (also the big-Q decode scratch)
eng.flow(img_a, img_b)                                           # ... and torch's allocator for this pair's map sizes
for tag, (ia, ib) in (('2x2 patch pairs (cathedral sizes)', (img_a, img_b)),):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = eng.flow(ia, ib)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    t = time.perf_counter()
    eng.flow(ia, ib, resample=False)
    torch.cuda.synchronize()
    dt_nr = time.perf_counter() - t
    pa, pb = eng._square_patches(ia), eng._square_patches(ib)
    boxes = np.array([[i[0], i[1], i[2], j[0], j[1], j[2]] for i in pa for j in pb], dtype=np.int32)
    crop = eng.make_cropper(ia, ib, torch.device('cuda'))
    buf = torch.empty((len(boxes), 3, 256, 512), device='cuda')
    q = torch.rand(len(boxes), 131072, 2, device='cuda')
    imgs = crop(boxes, buf)
    m(imgs, q)
    torch.cuda.synchronize()
    t = time.perf_counter()
    m(imgs, q)
    torch.cuda.synchronize()
    dm = time.perf_counter() - t
    print(f'dense pass, {tag}: {dt:.3f} s total ({dt_nr:.3f} s without the two visualisation warps, as gen_tasks calls it), of which the model call ({len(boxes)} x 131072 queries) {dm:.3f} s = '
          f'{len(boxes) * 131072 / dm:.0f} query-corr/s; the rest: crop launch, query grid upload, cotr_dense_cycle + '
          f'2 x cotr_dense_merge, D2H of the merged maps, the two visualisation warps', flush=True)
# dense pass + zoom, 10 k forced queries, converge_iters = 3, on a 512x512 synthetic pair
if '--config2' in sys.argv:
    ia, ib = synthetic_pair(4, (512, 512), (512, 512))
    rng = np.random.default_rng(1)
    q10k = np.stack([rng.uniform(5, 507, 10000), rng.uniform(5, 507, 10000)], 1)
    eng = ZoomEngine(m, max_pairs=1024)
    np.random.seed(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    corrs = eng.cotr_corr_multiscale(ia, ib, zooms, 3, max_corrs=10000, queries_a=q10k, force=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'config 2 (dense pass + 4 zoom levels, converge_iters 3, 10000 forced queries, 512x512 pair): {dt:.2f} s, '
          f'{len(corrs)} correspondences = {len(corrs) / dt:.0f} corr/s, {eng.total_tasks} crops through the model', flush=True)
    # the same call on the reference's FasterSparseEngine algorithm (squads of up to max_load + 1 queries per pilot crop), as
    # SURVEY 8(d) config 2 names it
    from cotr_amd.inference import FasterSparseEngine
    feng = FasterSparseEngine(m, 32, 'tile', max_load=256)
    np.random.seed(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fc = feng.cotr_corr_multiscale(ia, ib, zooms, 3, max_corrs=10000, queries_a=q10k, force=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f'config 2 on FasterSparseEngine(32, tile, max_load=256): {dt:.2f} s, {len(fc)} correspondences = {len(fc) / dt:.0f} corr/s '
          f'(with RANDOM weights the predictions are incoherent, squads of nearby tasks never form, and the reference algorithm - which '
          f'this class reproduces bit for bit on its goldens - leaves its grouped loop after one invocation per zoom level and its '
          f'roll-back loop only steps tasks already at the last zoom level: sparse_engine.py:383-410)', flush=True)
