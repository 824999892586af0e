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
from cotr_amd.build import LIB, LIB_EXP, build_library
from cotr_amd.models import build_model, NestedTensor
from cotr_amd.models.spec import state_spec
from cotr_amd.utils.synth import synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(experimental=False):
    """function names include/cotr_hip.h declares; the #ifdef COTR_EXPERIMENTAL block only for the experimental library"""
    src = open(os.path.join(ROOT, 'include', 'cotr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    if not experimental:
        src = re.sub(r'#ifdef COTR_EXPERIMENTAL.*?#endif', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cotr_[a-z0-9_]+)\s*\(', src)))


def exported_functions(path):
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if ' T cotr_' in l)


def dynamic_symbols(path):
    """every defined dynamic symbol of a shared object, whatever its type"""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.strip())


def test_library_builds_and_exports_every_declared_symbol():
    """Both ways round, for both libraries: every declared function is exported and every exported cotr_* function is declared
    (the experimental library = the product's declarations + the #ifdef COTR_EXPERIMENTAL block)."""
    build_library()
    build_library(experimental=True)
    for path, experimental in ((LIB, False), (LIB_EXP, True)):
        assert os.path.exists(path)
        names = declared_functions(experimental)
        assert len(names) >= 20
        assert exported_functions(path) == names, (set(names) ^ set(exported_functions(path)))
        # a sealed C ABI: the header's functions are the ONLY dynamic symbols - no kernel host stubs (_Z...), no thread-local knob /
        # device slots (cotr_tls_*), no C++ runtime leftovers (-fvisibility=hidden + a version script made from the header)
        extra = [n for n in dynamic_symbols(path) if n not in names]
        assert not extra, extra[:10]
    assert set(_lib.EXPORTED_SYMBOLS) <= set(declared_functions())          # the ctypes binding binds only declared symbols
    assert set(_lib.EXPERIMENTAL_SYMBOLS) == set(declared_functions(True)) - set(declared_functions())
    assert _lib.load_library().cotr_abi_version() == _lib.ABI_VERSION == 2
    src = open(os.path.join(ROOT, 'include', 'cotr_hip.h')).read()
    assert re.search(r'#define COTR_HIP_ABI_VERSION 2\b', src)
    # none of the former process-wide setters survives (ABI 2: knobs are per handle)
    assert not [n for n in exported_functions(LIB) if n.startswith('cotr_set_') and n not in
                ('cotr_set_knob', 'cotr_set_workspace', 'cotr_set_debug_taps', 'cotr_set_profiling')]


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
    """Knobs live in a registry per library handle (cotr_set_knob(h, ...)); handle NULL is the process-wide set the handle-less
    op-level entry points read.  count / name / get / set / reset need no device; the model object remembers knobs set before
    its handle exists.  The product library does not know the knobs of the measured dead ends."""
    k0 = _lib.knobs()
    assert len(k0) == 27 and all(cur == dflt for cur, dflt in k0.values())
    assert k0['attention_fusion_max_rows'] == (4096, 4096) and k0['encode_chunk'] == (64, 64)
    for exp_only in ('head_fusion_max_rows', 'ffn_preln', 'ffn_tail', 'coop_tail', 'coop_tail_spin', 'gemm_ln_min_rows', 'l2_warm', 'split_f16', 'split_f16_min_pairs'):
        assert exp_only not in k0
    try:
        _lib.set_knob('conv1x1_dense', 0)
        _lib.set_knob('ffn_fusion_max_rows', 0)
        k1 = _lib.knobs()
        assert k1['conv1x1_dense'] == (0, 1) and k1['ffn_fusion_max_rows'] == (0, 4096)
        with pytest.raises(_lib.CotrHipError):
            _lib.set_knob('no_such_knob', 1)
        with pytest.raises(_lib.CotrHipError):
            _lib.set_knob('coop_tail', 1)
        for name, bad in (('xcd_mapping', 3), ('xcd_mapping', 64), ('encode_chunk', 0), ('encode_chunk', 129), ('attention_splits', 3),
                          ('attention_fused_splits', 5), ('ks3', 2), ('attention_wide_occupancy', 4), ('ffn_fusion_max_rows', -1)):
            assert _lib.load_library().cotr_set_knob(None, name.encode(), bad) != 0, (name, bad)
        assert _lib.knobs() == k1                           # a refused value changes nothing
    finally:
        _lib.reset_knobs()
    assert _lib.knobs() == k0
    m = build_model(cotr_amd.default_args())                # no handle yet (CPU): remembered, applied when the handle is created
    m.set_knob('encode_chunk', 32)
    assert m.knobs()['encode_chunk'] == (32, 64) and _lib.knobs()['encode_chunk'] == (64, 64)
    m.reset_knobs()
    assert m.knobs()['encode_chunk'] == (64, 64)


def test_experimental_library_has_the_dead_ends_and_their_knobs():
    """libcotr_hip_exp.so (python -m cotr_amd.build --experimental): same ABI version, cotr_is_experimental() = 1, six more knobs
    (all at 'off'), two more op-level entry points.  Opened next to the product library with plain ctypes (RTLD_LOCAL)."""
    build_library(experimental=True)
    lib = ctypes.CDLL(LIB_EXP)
    assert lib.cotr_abi_version() == 2 and lib.cotr_is_experimental() == 1
    assert _lib.load_library().cotr_is_experimental() == 0
    lib.cotr_knob_name.restype = ctypes.c_char_p
    names = [lib.cotr_knob_name(i).decode() for i in range(lib.cotr_knob_count())]
    assert names[:27] == list(_lib.knobs()) and names[27:] == ['head_fusion_max_rows', 'ffn_preln', 'ffn_tail', 'coop_tail',
                                                                 'coop_tail_spin', 'gemm_ln_min_rows', 'l2_warm', 'split_f16', 'split_f16_min_pairs',
                                                                 'linear_rows_min_rows']
    for n, want in (('head_fusion_max_rows', 0), ('ffn_preln', 0), ('ffn_tail', 0), ('coop_tail', 0), ('gemm_ln_min_rows', 1 << 30),
                    ('l2_warm', 0), ('split_f16', 0), ('split_f16_min_pairs', 8)):
        cur, dflt = ctypes.c_int(), ctypes.c_int()
        assert lib.cotr_get_knob(None, n.encode(), ctypes.byref(cur), ctypes.byref(dflt)) == 0 and cur.value == dflt.value == want


def test_research_sources_are_derived_from_the_current_product_sources(tmp_path):
    """The research library has no forks of product files in the repository: csrc/experimental/patches/*.patch are applied to the CURRENT
    product sources at build time (cotr_amd/build.py generate_forks).  Every patch must fit (a product edit that breaks one fails here,
    before a stale research library can be compared against), the derived files keep every product line the patch does not touch, and no
    full copy of a product translation unit is tracked under csrc/experimental/."""
    import difflib
    from cotr_amd import build as b
    out = b.generate_forks(dest=str(tmp_path))
    for name, (src, _) in b.GENERATED.items():
        prod = open(os.path.join(b.CSRC, src)).read().splitlines()
        gen = open(os.path.join(out, name)).read().splitlines()
        sm = difflib.SequenceMatcher(None, prod, gen, autojunk=False)
        kept = sum(m.size for m in sm.get_matching_blocks())
        assert kept >= 0.9 * len(prod), f'{name}: only {kept} of {len(prod)} product lines survive the patch'
    tracked = os.listdir(os.path.join(b.CSRC, 'experimental'))
    assert not [f for f in b.GENERATED if f in tracked], 'a generated source is checked in next to its patch'


def test_first_run_kit_applies_the_one_line_switch(tmp_path):
    """tools/first_run_check.py --patch on a copy of the reference's COTR/models/__init__.py (authoring container only: the
    reference is not on the GPU box): the switch of INTEGRATION.md section 1 goes in, the original is kept, a second run is a
    no-op, and the patched module resolves build_model to the cotr_amd binding."""
    import importlib.util
    import shutil
    import subprocess
    import sys
    src = '/root/reference/COTR/models/__init__.py'
    if not os.path.exists(src):
        pytest.skip('no reference checkout here')
    pkg = tmp_path / 'COTR' / 'models'
    pkg.mkdir(parents=True)
    shutil.copy(src, pkg / '__init__.py')
    tool = os.path.join(ROOT, 'tools', 'first_run_check.py')
    for _ in range(2):
        subprocess.run([sys.executable, tool, '--patch', str(pkg / '__init__.py')], check=True, cwd=ROOT)
    assert (pkg / '__init__.py.orig').read_text() == open(src).read()
    patched = (pkg / '__init__.py').read_text()
    assert 'from cotr_amd.models import build_model' in patched and 'from .cotr_model import build' in patched
    spec = importlib.util.spec_from_file_location('patched_models', pkg / '__init__.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                    # cotr_amd is importable here: the try branch wins
    from cotr_amd.models import build_model as ours
    assert mod.build_model is ours


def test_encode_split_table_is_well_formed():
    """csrc/enc_split.inc (knob batch_split; written by tools/batch_cost.py from measured encode times): kEncFirst[n] is the first
    pass of n pairs - between 1 and n, and following it to the end covers n pairs in passes that are themselves unsplit first
    passes (what api.hip enc_next_chunk walks); the recorded times it was derived from are there for the reader."""
    import re
    src = open(os.path.join(os.path.dirname(LIB), 'enc_split.inc')).read()
    body = re.search(r'kEncFirst\[65\] = \{([^}]*)\}', src).group(1)
    first = [int(x) for x in body.split(',')]
    assert len(first) == 65 and first[0] == 0
    for n in range(1, 65):
        assert 1 <= first[n] <= n
        r, passes = n, []
        while r:
            passes.append(first[r])
            r -= first[r]
        assert sum(passes) == n and all(first[c] == c for c in passes[:1]), (n, passes)
    assert first[1] == 1 and first[16] == 16 and first[32] == 32 and first[64] == 64
