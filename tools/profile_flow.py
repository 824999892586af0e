"""Where the dense initial pass (ZoomEngine.flow = cotr_flow, inference_helper.py:168-182) spends its time, phase by phase, on the
cathedral demo pair's sizes (2 x 2 patch pairs x 131072 queries).  GPU box.    python tools/profile_flow.py [--resample]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cotr_amd
from cotr_amd.inference import ZoomEngine
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict
from tests.engine_fixtures import synthetic_pair

img_a, img_b = synthetic_pair(3, (783, 1064), (1053, 689))
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
eng = ZoomEngine(m)
resample = '--resample' in sys.argv
for _ in range(2):
    eng.flow(img_a, img_b, resample=resample)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter()
    eng.flow(img_a, img_b, resample=resample)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
print(f'flow(resample={resample}) on {img_a.shape[:2]} / {img_b.shape[:2]}: ' + ' '.join(f'{x * 1e3:.1f}' for x in ts) + ' ms per call')
# phase by phase (each phase synchronised: the sum is an upper bound of the call)
dev = torch.device('cuda')
pa, pb = eng._square_patches(img_a), eng._square_patches(img_b)
pairs = [(i, j) for i in pa for j in pb]
boxes = np.array([[i[0], i[1], i[2], j[0], j[1], j[2]] for i, j in pairs], dtype=np.int32)


def phase(name, fn, n=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print(f'  {name:58s} {(time.perf_counter() - t) / n * 1e3:8.2f} ms')
    return out


cropper = phase('cropper construction (image upload)', lambda: eng.make_cropper(img_a, img_b, dev))
buf = torch.empty((len(pairs), 3, 256, 512), dtype=torch.float32, device=dev)
img = phase('crop launch (4 patch pairs)', lambda: cropper(boxes, buf))


def grid():
    jj, ii = np.meshgrid(np.arange(512), np.arange(256))
    q_grid = np.stack([jj / 512, ii / 256], axis=-1)
    return torch.from_numpy(q_grid.reshape(1, -1, 2)).float().to(dev).expand(len(pairs), -1, -1).contiguous()


q = phase('query grid: numpy build + upload + expand', grid)
pred = phase('model call (4 x 131072 queries)', lambda: m(img, q)['pred_corrs'])
post = eng.make_dense_post(dev)
phase('dense post (cycle + 2 merges + D2H + float64)', lambda: post(pred, pairs, img_a.shape, img_b.shape))
if hasattr(post, 'device_maps'):
    phase('dense post, maps left on the device', lambda: post.device_maps(pred, pairs, img_a.shape, img_b.shape))
