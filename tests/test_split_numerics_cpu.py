"""The error model behind the RESEARCH knob split_f16 (docs/LABNOTES.md 3e, tools/split_mfma_numerics.py) pinned on the CPU: which split
low-precision scheme keeps the fp32-MFMA path's accuracy, and where its range ends.  numpy only - no GPU, no library."""
import importlib.util
import os

import numpy as np

_spec = importlib.util.spec_from_file_location(
    'split_mfma_numerics', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'split_mfma_numerics.py'))
N = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(N)


def _rms_rel(out, A, B):
    truth = A.astype(np.float64) @ B.astype(np.float64).T
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    return float(np.sqrt(np.mean(((out.astype(np.float64) - truth) / scale) ** 2)))


def test_f16_split_with_three_products_is_as_accurate_as_the_fp32_path():
    g = np.random.default_rng(3)
    for K in (256, 1152):
        A = np.maximum(g.standard_normal((24, K)), 0).astype(np.float32)
        B = (g.standard_normal((24, K)) / np.sqrt(K)).astype(np.float32)
        e32 = _rms_rel(N.gemm_terms([A], [B], [(0, 0)]), A, B)
        e16 = _rms_rel(N.gemm_f16x2(A, B, False), A, B)
        a2, b2 = N.split_bf16(A, 2), N.split_bf16(B, 2)
        ebf = _rms_rel(N.gemm_terms(a2, b2, [(0, 0), (0, 1), (1, 0)]), A, B)
        assert e16 <= 1.3 * e32, (K, e16, e32)          # the scheme of gemm_h2.h
        assert ebf >= 5 * e32, (K, ebf, e32)            # ... and why it is f16, not bf16, halves


def test_the_packed_form_is_hi_plus_lo_with_22_bits():
    g = np.random.default_rng(4)
    a = (g.standard_normal(4096) * np.exp(g.uniform(-4, 6, 4096))).astype(np.float32)     # 0.02 ... 400 in magnitude
    h, l = N.split_f16(a)
    back = h.astype(np.float64) + l.astype(np.float64) / 2048.0
    assert np.all(np.abs(back - a) <= np.abs(a) * 2.0 ** -22)
    assert np.array_equal((h + l / np.float32(2048.0)).astype(np.float32), back.astype(np.float32))   # unpacking is exact in fp32


def test_the_range_limits_are_where_the_header_says():
    g = np.random.default_rng(5)
    A = np.maximum(g.standard_normal((16, 256)), 0).astype(np.float32)
    B = (g.standard_normal((16, 256)) / 16).astype(np.float32)
    e32 = _rms_rel(N.gemm_terms([A], [B], [(0, 0)]), A, B)
    tiny = (B * np.float32(1e-5)).astype(np.float32)                  # a tensor whose whole scale sits in f16's subnormals
    assert _rms_rel(N.gemm_f16x2(A, tiny, False), A, tiny) > 20 * e32
    with np.errstate(all='ignore'):
        big = N.gemm_f16x2((A * np.float32(1e5)).astype(np.float32), B, False)   # above 65504
    assert not np.isfinite(big).all()
