"""Restatement of Pillow's 8-bit bilinear resample (third party; the reference calls
``PIL.Image.fromarray(patch).resize((256, 256), resample=PIL.Image.BILINEAR)`` at
``COTR/inference/refinement_task.py:117-118`` and ``inference_helper.py:110-111,190-191``).

Pillow is pinned by the reference at ``environment.yml`` (pillow=8.x); the algorithm restated here is
``ImagingResample`` of ``src/libImaging/Resample.c`` (unchanged since Pillow 3.4 for 8-bit images):
separable two-pass (horizontal first), double-precision triangle-filter coefficients whose support grows
with the down-scale factor, coefficients quantised to 22 fractional bits, 32-bit integer accumulation
with a rounding bias, and the horizontal result rounded back to uint8 before the vertical pass.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker for the HIP crop+resize kernel
(``cotr_amd/csrc/crop_resize.hip``).  It is itself checked bit-for-bit against the installed Pillow in
``tests/test_crop_resize_cpu.py``.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc (Resample.c) for box = (0, in_size), bilinear."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, dtype=np.float64)
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0 else a
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = float(sum(w))  # C accumulates left to right in double
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_size=256):
    """img: uint8 [H, W, C] -> uint8 [out_size, out_size, C], bit-identical to
    ``np.array(PIL.Image.fromarray(img).resize((out_size, out_size), resample=PIL.Image.BILINEAR))``."""
    h, w, c = img.shape
    src = img.astype(np.int64)
    if w != out_size:
        bounds, kk = _coeffs(w, out_size)
        tmp = np.empty((h, out_size, c), dtype=np.uint8)
        for xx in range(out_size):
            xmin, xmax = bounds[xx]
            acc = (src[:, xmin:xmin + xmax, :] * kk[xx, :xmax][None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        src = tmp.astype(np.int64)
    if h != out_size:
        bounds, kk = _coeffs(h, out_size)
        out = np.empty((out_size, src.shape[1], c), dtype=np.uint8)
        for yy in range(out_size):
            ymin, ymax = bounds[yy]
            acc = (src[ymin:ymin + ymax, :, :] * kk[yy, :ymax][:, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        return out
    return src.astype(np.uint8)


def crop_resize_normalize(img_from, img_to, box_from, box_to):
    """One engine task's network input (refinement_task.py:105-120): square crops (x, y, size) of both images ->
    256x256 bilinear -> side by side -> to_tensor -> ImageNet normalise.  Returns float32 [3,256,512]."""
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    halves = []
    for img, (x, y, size) in ((img_from, box_from), (img_to, box_to)):
        halves.append(resize_bilinear_u8(img[y:y + size, x:x + size]))
    canvas = np.concatenate(halves, axis=1)                       # [256, 512, 3]
    t = canvas.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return ((t - mean[:, None, None]) / std[:, None, None]).astype(np.float32)
