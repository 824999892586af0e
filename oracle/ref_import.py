"""Import the reference's own ``COTR.models`` UNCHANGED from ``/root/reference``.

Authoring-container only (``/root/reference`` is not on the GPU box).  The
reference cannot be imported as-is here because ``torchvision`` and ``cv2`` are
not installed; this module puts minimal stub modules in
``sys.modules`` so that ``import COTR.models`` succeeds and every line of the
reference's model code runs untouched on torch CPU:

* ``torchvision``: ``__version__='0.8.2'``, ``_is_tracing()``, ``models.resnet50``
  and ``models._utils.IntermediateLayerGetter`` (restated in
  ``oracle/resnet_body.py``), ``transforms.functional.{to_tensor,normalize}``.
* ``cv2`` / ``imageio`` / ``tables``: empty modules (only touched by
  visualisation / dataset helpers that the forward path never calls).

Used by ``tests/golden/make_golden.py`` to generate the committed golden vectors
and by ``tests/test_oracle_golden.py::test_restatement_matches_live_reference``
(skipped when ``/root/reference`` is absent).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).
"""
import argparse
import contextlib
import io
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'COTR', 'models'))


def _install_stubs():
    import numpy as np
    import torch
    from . import resnet_body

    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        tv.__version__ = '0.8.2'
        tv._is_tracing = lambda: False
        models = types.ModuleType('torchvision.models')

        def resnet50(pretrained=False, norm_layer=None, replace_stride_with_dilation=None, **kw):
            # pretrained ImageNet weights are a download: ignored.
            return resnet_body.ResNet50(norm_layer, replace_stride_with_dilation)

        models.resnet50 = resnet50
        _utils = types.ModuleType('torchvision.models._utils')
        _utils.IntermediateLayerGetter = resnet_body.IntermediateLayerGetter
        models._utils = _utils
        transforms = types.ModuleType('torchvision.transforms')
        functional = types.ModuleType('torchvision.transforms.functional')

        def to_tensor(pic):
            arr = np.asarray(pic)
            t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
            return t.float().div(255) if arr.dtype == np.uint8 else t

        def normalize(t, mean, std):
            mean = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
            return (t - mean) / std

        functional.to_tensor = to_tensor
        functional.normalize = normalize
        transforms.functional = functional
        tv.models = models
        tv.transforms = transforms
        sys.modules.update({
            'torchvision': tv,
            'torchvision.models': models,
            'torchvision.models._utils': _utils,
            'torchvision.transforms': transforms,
            'torchvision.transforms.functional': functional,
        })
    for name in ('cv2', 'imageio', 'tables'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod = types.ModuleType(name)
                if name == 'cv2':
                    mod.INTER_LINEAR = 1

                    def getAffineTransform(src, dst):
                        """2x3 affine map through 3 point pairs (what cv2.getAffineTransform solves), float64;
                        the reference needs it at COTR/inference/inference_helper.py:155-156."""
                        src = np.asarray(src, dtype=np.float64)
                        dst = np.asarray(dst, dtype=np.float64)
                        a = np.concatenate([src, np.ones((3, 1))], axis=1)
                        return np.linalg.solve(a, dst).T

                    mod.getAffineTransform = getAffineTransform
                sys.modules[name] = mod
    if not hasattr(np, 'int'):
        np.int = int  # removed in numpy 2.x; used at COTR/inference/sparse_engine.py:171


def default_args(**over):
    """The argparse defaults of COTR/options/options.py:41-51 + demo_single_pair.py:58-62."""
    a = argparse.Namespace(backbone='resnet50', hidden_dim=256, dilation=False, dropout=0.1,
                           nheads=8, layer='layer3', enc_layers=6, dec_layers=6,
                           position_embedding='lin_sine', dim_feedforward=1024)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def import_reference_models():
    """Returns the reference's ``COTR.models`` module (imported from /root/reference)."""
    if not reference_available():
        raise RuntimeError(f'{REFERENCE_ROOT} is not present on this machine')
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import COTR.models as ref_models  # noqa: E402
    return ref_models


def build_reference_model(args=None, quiet=True):
    """``COTR.models.build_model(args)`` of the reference, in eval mode on CPU."""
    ref_models = import_reference_models()
    args = args or default_args()
    ctx = contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()
    with ctx:
        model = ref_models.build_model(args)
    return model.eval()
