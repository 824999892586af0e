"""Seeded synthetic weights and inputs (numpy PCG64, platform independent).

No trained COTR checkpoint nor ImageNet ResNet-50 weights exist offline
(SURVEY.md fact 0.5), so the benchmark and the parity tests run on random
weights.  The distributions follow the reference's initialisers (kaiming
fan-out for convs as torchvision does, xavier-uniform for every transformer
matrix as ``COTR/models/transformer.py:42-45`` does, torch's Linear default for
``corr_embed``) but ALSO randomise every bias, every LayerNorm affine and all
four FrozenBN buffers, so that a kernel which drops or misplaces any of them
fails parity.  The last BN of each bottleneck gets a smaller gain so the
residual stream stays O(1) like in a trained network.
"""
import math
from collections import OrderedDict

import numpy as np

from ..models.spec import state_spec


def synth_state_dict(seed=0, attn_gain=1.0, as_torch=True, **spec_kw):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, (shape, kind) in state_spec(**spec_kw).items():
        if kind == 'conv':
            fan_out = shape[0] * shape[2] * shape[3]
            w = rng.standard_normal(shape) * math.sqrt(2.0 / fan_out)
        elif kind in ('mat', 'mlp_w'):
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            bound = math.sqrt(6.0 / (fan_in + fan_out)) if kind == 'mat' else 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-bound, bound, shape)
            if name.endswith('in_proj_weight') and attn_gain != 1.0:
                w[: 2 * shape[1]] *= attn_gain  # sharpen q and k -> peakier softmax
        elif kind == 'bias':
            w = 0.05 * rng.standard_normal(shape)
        elif kind == 'mlp_b':
            w = rng.uniform(-1.0 / 16.0, 1.0 / 16.0, shape)
        elif kind == 'ln_w':
            w = rng.uniform(0.8, 1.2, shape)
        elif kind == 'ln_b':
            w = 0.05 * rng.standard_normal(shape)
        elif kind == 'bn_w':
            w = rng.uniform(0.8, 1.2, shape)
        elif kind == 'bn_w_last':
            w = rng.uniform(0.15, 0.35, shape)
        elif kind == 'bn_b':
            w = 0.1 * rng.standard_normal(shape)
        elif kind == 'bn_rm':
            w = 0.1 * rng.standard_normal(shape)
        elif kind == 'bn_rv':
            w = rng.uniform(0.75, 1.25, shape)
        else:
            raise AssertionError(kind)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    if as_torch:
        import torch
        return OrderedDict((k, torch.from_numpy(v)) for k, v in out.items())
    return out


def synth_inputs(batch, queries, seed=1, as_torch=True):
    """img ~ N(0,1) [B,3,256,512] (ImageNet-normalised pixels have that range),
    queries ~ U[0,1)^2 [B,Q,2] (x<0.5: left image, x>=0.5: right image)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.standard_normal((batch, 3, 256, 512)).astype(np.float32)
    q = rng.random((batch, queries, 2)).astype(np.float32)
    if as_torch:
        import torch
        return torch.from_numpy(img), torch.from_numpy(q)
    return img, q


def state_checksum(sd):
    """float64 (sum, sum of squares) over all tensors, to detect generator drift."""
    s = s2 = 0.0
    for v in sd.values():
        a = np.asarray(v, dtype=np.float64)
        s += float(a.sum())
        s2 += float((a * a).sum())
    return s, s2
