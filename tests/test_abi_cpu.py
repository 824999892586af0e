"""CPU-side checks of the boundary: the C-ABI library builds for gfx950, loads, and exports exactly
the entry points include/cotr_hip.h declares; the model object keeps the reference's state-dict and
attribute contract and refuses to run without the GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

import cotr_amd
from cotr_amd import _lib
from cotr_amd.build import LIB, build_library
from cotr_amd.models import build_model, NestedTensor
from cotr_amd.models.spec import state_spec
from cotr_amd.utils.synth import synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'cotr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cotr_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    build_library()
    assert os.path.exists(LIB)
    lib = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/cotr_hip.h but not exported'
    assert set(_lib.EXPORTED_SYMBOLS) <= set(names)          # the ctypes binding binds only declared symbols
    assert _lib.load_library().cotr_abi_version() == 1


def test_error_paths_without_a_gpu():
    lib = _lib.load_library()
    h = ctypes.c_void_p()
    if not torch.cuda.is_available():
        assert lib.cotr_create(ctypes.byref(h), 0) != 0       # no device: an error code, not a crash
        assert lib.cotr_last_error(None)
    assert lib.cotr_encode(None, None, 1, None) == -1         # COTR_ERR_ARG on a null handle
    assert lib.cotr_decode(None, None, 1, 1, None, None) == -1


def test_state_dict_contract_matches_the_reference_layout():
    m = build_model(cotr_amd.default_args())
    spec = state_spec()
    sd = m.state_dict()
    assert list(sorted(sd)) == list(sorted(spec))
    assert all(tuple(sd[k].shape) == spec[k][0] for k in spec)
    assert sum(v.numel() for v in sd.values()) == 18449090    # SURVEY.md 2.2 [probe]
    # attribute groups train_cotr.py:49-55 builds its optimiser from
    for attr in ('transformer', 'corr_embed', 'query_proj', 'input_proj', 'backbone'):
        assert hasattr(m, attr) and list(getattr(m, attr).parameters()) is not None
    assert next(m.parameters()).device.type == 'cpu'          # engines discover the device this way
    # torchvision checkpoints carry num_batches_tracked; the reference drops it on load (backbone.py:36-44)
    sd2 = synth_state_dict(0)
    sd2['backbone.0.body.bn1.num_batches_tracked'] = torch.tensor(0)
    m.load_state_dict(sd2)
    # utils.safe_load_weights retries with 'module.' stripped (COTR/utils/utils.py:164-193)
    with pytest.raises(RuntimeError):
        m.load_state_dict({'module.' + k: v for k, v in synth_state_dict(0).items()})


def test_no_cpu_fallback_and_shape_contract():
    m = build_model(cotr_amd.default_args()).eval()
    with pytest.raises(_lib.CotrHipError):
        m(torch.zeros(1, 3, 256, 512), torch.zeros(1, 4, 2))
    with pytest.raises(AssertionError):                       # COTR/models/backbone.py:80
        m(torch.zeros(1, 3, 256, 500), torch.zeros(1, 4, 2))
    with pytest.raises(_lib.CotrHipError):                    # training mode: no CPU fallback either
        m.train()(torch.zeros(1, 3, 256, 512), torch.zeros(1, 4, 2))
    with pytest.raises(NotImplementedError):                  # encode()/decode() are the inference split
        m.train().encode(torch.zeros(1, 3, 256, 512))
    with pytest.raises(_lib.CotrHipError):                    # trainable backbone (stages 2-3): GPU only as well
        m2 = build_model(cotr_amd.default_args(lr_backbone=1e-5)).train()
        m2(torch.zeros(1, 3, 256, 512), torch.zeros(1, 4, 2))
    with pytest.raises(NotImplementedError):
        build_model(cotr_amd.default_args(layer='layer2', dim_feedforward=512))
    assert isinstance(NestedTensor(torch.zeros(1, 3, 256, 512), None).decompose()[0], torch.Tensor)
    # a real key-padding mask is refused (no masked attention in the HIP path), an all-False one is what the reference feeds
    from cotr_amd.models.cotr_model import COTR
    img = torch.zeros(1, 3, 256, 512)
    assert COTR._as_batch(NestedTensor(img, torch.zeros(1, 256, 512, dtype=torch.bool))) is img
    mask = torch.zeros(1, 256, 512, dtype=torch.bool)
    mask[0, 0, 0] = True
    with pytest.raises(NotImplementedError):
        COTR._as_batch(NestedTensor(img, mask))


def test_knob_registry_round_trip_without_a_gpu():
    """cotr_set_* switches are process-wide; the registry (cotr_knob_count / _name / get / set / reset) is what tests and A/B
    tools snapshot and restore through.  Setting and resetting needs no device."""
    k0 = _lib.knobs()
    assert len(k0) >= 17 and all(cur == dflt for cur, dflt in k0.values())
    assert k0['head_fusion_max_rows'] == (0, 0) and k0['attention_fusion_max_rows'] == (1024, 1024)
    try:
        _lib.set_knob('head_fusion_max_rows', 2048)
        _lib.set_knob('conv1x1_dense', 0)
        assert _lib.load_library().cotr_set_ffn_fusion_max_rows(0) == 0      # the direct setters record too
        k1 = _lib.knobs()
        assert k1['head_fusion_max_rows'] == (2048, 0) and k1['conv1x1_dense'] == (0, 1) and k1['ffn_fusion_max_rows'] == (0, 1024)
        with pytest.raises(_lib.CotrHipError):
            _lib.set_knob('no_such_knob', 1)
        assert _lib.load_library().cotr_set_knob(b'xcd_mapping', 3) != 0    # the setter's own range check applies
    finally:
        _lib.reset_knobs()
    assert _lib.knobs() == k0
