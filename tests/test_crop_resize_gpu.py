"""HIP crop + Pillow-bilinear resize + normalise kernel against Pillow itself (and torch's to_tensor/normalize
arithmetic) on the MI355X: bit-exact."""
import ctypes

import numpy as np
import PIL.Image
import pytest
import torch

from cotr_amd import _lib

pytestmark = pytest.mark.gpu


def reference(img_a, img_b, boxes):
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    out = []
    for xa, ya, sa, xb, yb, sb in boxes:
        ha = np.array(PIL.Image.fromarray(img_a[ya:ya + sa, xa:xa + sa]).resize((256, 256), resample=PIL.Image.BILINEAR))
        hb = np.array(PIL.Image.fromarray(img_b[yb:yb + sb, xb:xb + sb]).resize((256, 256), resample=PIL.Image.BILINEAR))
        canvas = np.concatenate([ha, hb], axis=1)
        t = torch.from_numpy(canvas.transpose(2, 0, 1).copy()).float().div(255)
        out.append((t - mean) / std)
    return torch.stack(out)


def run_kernel(img_a, img_b, boxes):
    lib = _lib.load_library()
    d = torch.device('cuda:0')
    ta, tb = torch.from_numpy(img_a).to(d), torch.from_numpy(img_b).to(d)
    bx = torch.tensor(boxes, dtype=torch.int32, device=d)
    out = torch.empty(len(boxes), 3, 256, 512, device=d)
    max_size = max(max(b[2], b[5]) for b in boxes)
    rc = lib.cotr_crop_resize_pairs(ctypes.c_void_p(ta.data_ptr()), img_a.shape[0], img_a.shape[1],
                                    ctypes.c_void_p(tb.data_ptr()), img_b.shape[0], img_b.shape[1],
                                    ctypes.c_void_p(bx.data_ptr()), len(boxes), ctypes.c_void_p(out.data_ptr()), max_size,
                                    _lib.current_stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize('shape_a,shape_b,sizes', [
    ((300, 420), (350, 330), [(150, 222), (300, 330), (18, 20), (256, 256), (74, 300)]),   # up- and down-scaling
    ((1064, 783), (689, 1053), [(390, 344), (782, 688), (48, 42), (196, 172)]),            # cathedral demo sizes, zooms 1/2..1/16
    ((2100, 2048), (64, 64), [(2048, 64), (1024, 2), (4, 32)]),                             # 8x down-scale, tiny crops
])
def test_bit_exact_against_pillow(shape_a, shape_b, sizes):
    rng = np.random.default_rng(sum(shape_a) + sum(shape_b))
    img_a = rng.integers(0, 256, shape_a + (3,), dtype=np.uint8)
    img_b = rng.integers(0, 256, shape_b + (3,), dtype=np.uint8)
    boxes = []
    for sa, sb in sizes:
        for _ in range(3):
            boxes.append((int(rng.integers(0, shape_a[1] - sa + 1)), int(rng.integers(0, shape_a[0] - sa + 1)), sa,
                          int(rng.integers(0, shape_b[1] - sb + 1)), int(rng.integers(0, shape_b[0] - sb + 1)), sb))
    out = run_kernel(img_a, img_b, boxes)
    ref = reference(img_a, img_b, boxes)
    assert torch.equal(out, ref), float((out - ref).abs().max())


def test_smooth_image_and_borders():
    yy, xx = np.mgrid[0:500, 0:640]
    img = np.stack([(xx * 255 / 639), (yy * 255 / 499), ((xx + yy) % 256)], -1).astype(np.uint8)
    boxes = [(0, 0, 500, 140, 0, 500), (640 - 62, 500 - 62, 62, 0, 500 - 124, 124)]
    assert torch.equal(run_kernel(img, img, boxes), reference(img, img, boxes))
