"""Deterministic stand-ins used to pin the engine's STATE MACHINE (not the network): a synthetic image pair and a
fake 'model' whose answer is a smooth per-sample function of the crop and the query, evaluated sample by sample in
float64 numpy so it does not depend on how crops are batched.  Test infrastructure."""
import hashlib

import numpy as np
import torch


def digest(arr):
    """sha256 of an array's dtype, shape and bytes: bit-exact comparison of whole maps without storing them."""
    a = np.ascontiguousarray(arr)
    return hashlib.sha256(str((a.dtype.str, a.shape)).encode() + a.tobytes()).digest()


def ids(idx):
    """identifiers as the goldens store them: None (a task made without identifier, as in the reference) -> -1"""
    return np.array([-1 if i is None else i for i in idx], dtype=np.int64)


def synthetic_pair(seed=0, shape_a=(300, 420), shape_b=(350, 330)):
    rng = np.random.default_rng(seed)

    def tex(h, w):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = np.zeros((h, w, 3))
        for c in range(3):
            for _ in range(6):
                fx, fy, ph = rng.uniform(0.01, 0.15), rng.uniform(0.01, 0.15), rng.uniform(0, 6.28)
                img[..., c] += np.sin(fx * xx + fy * yy + ph)
        img = (img - img.min()) / (img.max() - img.min())
        return (img * 255).astype(np.uint8)
    return tex(*shape_a), tex(*shape_b)


class FakeModel(torch.nn.Module):
    """model(img[B,3,256,512], q[B,Q,2]) -> {'pred_corrs': [B,Q,2]} with x in the right half."""

    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.calls = []
        xs = np.linspace(-1, 1, 256)
        self.ramp_x = np.broadcast_to(xs[None, None, :], (3, 256, 256))
        self.ramp_y = np.broadcast_to(xs[None, :, None], (3, 256, 256))

    def forward(self, img, queries):
        self.calls.append((tuple(img.shape), tuple(queries.shape)))
        im = img.detach().cpu().numpy().astype(np.float64)
        qs = queries.detach().cpu().numpy().astype(np.float64)
        out = np.zeros(qs.shape, dtype=np.float32)
        for b in range(im.shape[0]):
            right, left = im[b, :, :, 256:], im[b, :, :, :256]
            wx = float(np.mean(right * self.ramp_x)) - 0.5 * float(np.mean(left * self.ramp_x))
            wy = float(np.mean(right * self.ramp_y)) - 0.5 * float(np.mean(left * self.ramp_y))
            out[b, :, 0] = 0.75 + 0.2 * np.tanh(3 * wx + 0.6 * (qs[b, :, 0] - 0.25))
            out[b, :, 1] = 0.5 + 0.4 * np.tanh(3 * wy + 0.6 * (qs[b, :, 1] - 0.5))
        return {'pred_corrs': torch.from_numpy(out).to(img.device)}


def pil_cropper_factory(img_a, img_b, device):
    """Host-side cropper with the reference's own recipe (refinement_task.py:105-120), for CPU tests of the engine
    logic; the product default is the HIP kernel."""
    import PIL.Image
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def crop(boxes, out):
        for i, (xa, ya, sa, xb, yb, sb) in enumerate(np.asarray(boxes)):
            ha = np.array(PIL.Image.fromarray(img_a[ya:ya + sa, xa:xa + sa]).resize((256, 256), resample=PIL.Image.BILINEAR))
            hb = np.array(PIL.Image.fromarray(img_b[yb:yb + sb, xb:xb + sb]).resize((256, 256), resample=PIL.Image.BILINEAR))
            canvas = np.concatenate([ha, hb], axis=1)
            t = torch.from_numpy(canvas.transpose(2, 0, 1).copy()).float().div(255)
            out[i] = ((t - mean) / std).to(out.device)
        return out[:len(boxes)]
    return crop


class CyclicFakeModel(FakeModel):
    """Approximately cycle-consistent stand-in (left-half queries land in the right half and vice versa, with a small
    smooth, image-dependent displacement), so that the reference's dense pass (cotr_flow) yields confident pixels."""

    def forward(self, img, queries):
        self.calls.append((tuple(img.shape), tuple(queries.shape)))
        im = img.detach().cpu().numpy().astype(np.float64)
        qs = queries.detach().cpu().numpy().astype(np.float64)
        out = np.zeros(qs.shape, dtype=np.float32)
        for b in range(im.shape[0]):
            right, left = im[b, :, :, 256:], im[b, :, :, :256]
            c = float(np.tanh(np.mean(right * self.ramp_x) - np.mean(left * self.ramp_x)))
            x, y = qs[b, :, 0], qs[b, :, 1]
            is_left = x < 0.5
            xl = np.where(is_left, x, x - 0.5)                      # position inside the own half
            dx = 0.01 * np.sin(2 * np.pi * y + c)
            dy = 0.01 * np.sin(4 * np.pi * xl + c)
            out[b, :, 0] = np.where(is_left, x + 0.5 + dx, x - 0.5 - dx)
            out[b, :, 1] = np.where(is_left, y + dy, y - dy)
        return {'pred_corrs': torch.from_numpy(out).to(img.device)}
