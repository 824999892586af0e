#!/usr/bin/env python3
"""Error analysis for "fp32 products from split low-precision MFMAs" (round-3 verdict item 8) - CPU only, numpy.

The question: gfx950 runs bf16 / f16 MFMAs at 16x the rate of v_mfma_f32_32x32x2_f32.  If an fp32 operand is split into a few
16-bit terms, a dot product of fp32 numbers becomes a handful of 16-bit MFMAs accumulating in fp32.  How far is each such scheme
from the fp32-MFMA path this package ships, measured against the fp64 truth, on operands shaped like this model's?

Schemes (all accumulate in fp32, one rounding per added product, K in order - a pessimistic model of the matrix pipe, the same for all):
  f32            a*b exact (fma), the shipped path
  bf16x3 / 6     a = a0+a1+a2 (bf16 each, 24 significand bits together), products a0b0 a0b1 a1b0 a1b1 a0b2 a2b0          2.7x theoretical
  bf16x3 / 9     all nine cross terms                                                                                   1.8x
  bf16x2 / 3     a = a0+a1 (16 bits), products a0b0 a0b1 a1b0                                                              5.3x
  f16x2 / 3      a = h + l*2^-11 (h = f16(a), l = f16((a-h)*2^11): 22 bits), products hh, hl + lh in a second accumulator     5.3x
  f16x2 / 4      the same with l*l kept                                                                                 4x
Operands: (1) a GEMM of the forward: post-ReLU activations x N(0, 1/sqrt(K)) weights, K = 256 / 1024 / 2304; (2) the attention logits
of the peaky goldens: q.k with |logit| up to ~200 before the softmax (a relative error of 1e-6 in a logit of 200 is 2e-4 in exp).

usage: python tools/split_mfma_numerics.py [--rows 48 --cols 48]
"""
import argparse
import numpy as np


def to_bf16(x):
    """round-to-nearest-even to bf16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def to_f16(x):
    with np.errstate(over='ignore'):
        return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def split_bf16(x, n):
    parts, r = [], np.asarray(x, np.float32).copy()
    for _ in range(n):
        p = to_bf16(r)
        parts.append(p)
        r = (r - p).astype(np.float32)          # exact in fp32
    return parts


def split_f16(x):
    h = to_f16(x)
    lo = to_f16((np.asarray(x, np.float32) - h) * np.float32(2048.0))
    return h, lo


def fma_acc(acc, a, b):
    """acc + a*b with ONE fp32 rounding (the product of two <=24-bit numbers is exact in fp64)"""
    return (acc.astype(np.float64) + a.astype(np.float64) * b.astype(np.float64)).astype(np.float32)


def gemm_terms(A, B, terms):
    """sum_k sum_(i,j in terms) A_i[:,k] * B_j[:,k] accumulated in fp32 in that order.  A_i: [M,K], B_j: [N,K]"""
    M, K = A[0].shape
    N = B[0].shape[0]
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        for i, j in terms:
            acc = fma_acc(acc, A[i][:, k][:, None], B[j][:, k][None, :])
    return acc


def gemm_f16x2(A, B, keep_ll):
    h_a, l_a = split_f16(A)
    h_b, l_b = split_f16(B)
    M, K = A.shape
    N = B.shape[0]
    main = np.zeros((M, N), np.float32)
    cross = np.zeros((M, N), np.float32)
    tiny = np.zeros((M, N), np.float32)
    for k in range(K):
        main = fma_acc(main, h_a[:, k][:, None], h_b[:, k][None, :])
        cross = fma_acc(cross, h_a[:, k][:, None], l_b[:, k][None, :])
        cross = fma_acc(cross, l_a[:, k][:, None], h_b[:, k][None, :])
        if keep_ll:
            tiny = fma_acc(tiny, l_a[:, k][:, None], l_b[:, k][None, :])
    out = main + cross * np.float32(2.0 ** -11)
    if keep_ll:
        out = out + tiny * np.float32(2.0 ** -22)
    return out.astype(np.float32)


def report(name, A, B):
    truth = A.astype(np.float64) @ B.astype(np.float64).T
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T      # sum |a||b|: the condition-free yardstick
    res = {}
    res['f32'] = gemm_terms([A], [B], [(0, 0)])
    a3, b3 = split_bf16(A, 3), split_bf16(B, 3)
    res['bf16x3 / 6'] = gemm_terms(a3, b3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])
    res['bf16x3 / 9'] = gemm_terms(a3, b3, [(i, j) for i in range(3) for j in range(3)])
    res['bf16x2 / 3'] = gemm_terms(a3[:2], b3[:2], [(0, 0), (0, 1), (1, 0)])
    res['f16x2 / 3'] = gemm_f16x2(A, B, False)
    res['f16x2 / 4'] = gemm_f16x2(A, B, True)
    print(f'{name}: K = {A.shape[1]}, {A.shape[0]} x {B.shape[0]} outputs, |truth| median {np.median(np.abs(truth)):.3g}, max {np.abs(truth).max():.3g}')
    base = None
    for k, v in res.items():
        err = np.abs(v.astype(np.float64) - truth)
        rel = err / scale
        rms = float(np.sqrt(np.mean(rel ** 2)))
        if base is None:
            base = rms
        print(f'    {k:12s} error / sum|a||b|: rms {rms:9.3g}  max {rel.max():9.3g}   = {rms / base:7.1f} x the f32 path'
              f'   max abs error {err.max():9.3g}' + ('   (overflow)' if not np.isfinite(v).all() else ''))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=48)
    ap.add_argument('--cols', type=int, default=48)
    a = ap.parse_args()
    g = np.random.default_rng(0)
    for K in (256, 1024, 2304):
        A = np.maximum(g.standard_normal((a.rows, K)), 0).astype(np.float32)               # post-ReLU activations
        B = (g.standard_normal((a.cols, K)) / np.sqrt(K)).astype(np.float32)               # weights
        report('conv / linear', A, B)
    # attention logits of the peaky goldens: per head 32 channels, q scaled so that logits reach ~ +-200
    q = (g.standard_normal((a.rows, 32)) * 6).astype(np.float32)
    k = (g.standard_normal((a.cols, 32)) * 6).astype(np.float32)
    report('attention logits (peaky: gain 32^2)', q, k)
    # a tensor whose WHOLE scale is tiny: below f16's normal range (6.1e-5) the hi term loses bits that the lo term cannot give back
    A = np.maximum(g.standard_normal((a.rows, 256)), 0).astype(np.float32)
    for sc in (1e-3, 1e-5, 1e-6):
        B = (g.standard_normal((a.cols, 256)) / 16 * sc).astype(np.float32)
        report(f'weights of scale {sc:g} (f16 subnormals below 6.1e-5)', A, B)
    # large activations: f16 overflows above 65504
    A = (np.maximum(g.standard_normal((a.rows, 256)), 0) * 1e5).astype(np.float32)
    B = (g.standard_normal((a.cols, 256)) / 16).astype(np.float32)
    report('activations of 1e5 (f16 range)', A, B)


if __name__ == '__main__':
    main()
