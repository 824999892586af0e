"""Op-level parity of the training-step kernels (cotr_amd/csrc/train.hip, attention_train.hip) through their autograd
wrappers (cotr_amd/train_ops.py): forward values and every gradient against the same op written with torch in fp64 on the
CPU.  Dropout is 0 here (exact comparison); the masks' forward / backward consistency is pinned end to end by
tests/test_training_gpu.py::test_dropout_masks_are_consistent_between_forward_and_backward."""
import math

import pytest
import torch
import torch.nn.functional as F

from cotr_amd import train_ops as T

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _leaf(t):
    return t.cuda().requires_grad_()


def _leaf64(t):
    return t.double().requires_grad_()


@pytest.mark.parametrize('rows', [1000, 37, 8192])
@pytest.mark.parametrize('with_x', [True, False])
def test_residual_layernorm_forward_backward(rows, with_x):
    g = _g(rows)
    x, a = torch.randn(rows, 256, generator=g), torch.randn(rows, 256, generator=g) * 2 + 0.3
    w, b = torch.rand(256, generator=g) + 0.5, 0.1 * torch.randn(256, generator=g)
    dy = torch.randn(rows, 256, generator=g)
    xs = [_leaf(x) if with_x else None, _leaf(a), _leaf(w), _leaf(b)]
    y = T.AddDropLN.apply(xs[0], xs[1], xs[2], xs[3], 0.0)
    grads = torch.autograd.grad(y, [t for t in xs if t is not None], dy.cuda())
    rs = [_leaf64(x) if with_x else None, _leaf64(a), _leaf64(w), _leaf64(b)]
    s = rs[1] + rs[0] if with_x else rs[1]
    yr = F.layer_norm(s, (256,), rs[2], rs[3])
    refs = torch.autograd.grad(yr, [t for t in rs if t is not None], dy.double())
    assert _rel(y, yr) < 1e-5
    for got, want in zip(grads, refs):
        assert _rel(got, want) < 3e-5


@pytest.mark.parametrize('M,N,K,relu', [(1000, 256, 256, False), (8192, 1024, 256, True), (48, 256, 1024, False),
                                        (1030, 512, 256, True), (200, 256, 1024, False),
                                        (2100, 256, 1024, False), (4096, 512, 256, True), (16384, 1024, 256, True)])   # (the 128 x 128 TN kernel)
def test_linear_forward_backward(M, N, K, relu):
    """Proj with one slice = nn.Linear (+ ReLU): dX on the cached W^T, dW by the transpose-free split-M kernel, db column sums."""
    g = _g(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    b, dy = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    xs = [_leaf(x), _leaf(w), _leaf(b)]
    y = T.linear(xs[0], xs[1], xs[2], relu=relu)
    grads = torch.autograd.grad(y, xs, dy.cuda())
    rs = [_leaf64(x), _leaf64(w), _leaf64(b)]
    yr = F.linear(*rs)
    # the ReLU mask of the reference is the kernel's own (y > 0): an element within rounding of zero may fall on either side
    # in fp32 vs fp64, and then carries a whole dy - not an error of the backward kernels
    yr = yr * (y.detach().cpu() > 0).double() if relu else yr
    refs = torch.autograd.grad(yr, rs, dy.double())
    assert _rel(y, yr) < 2e-5
    for got, want in zip(grads, refs):
        assert _rel(got, want) < 5e-5
    # the W^T cache follows the parameter: an in-place update (optimiser step) invalidates it
    with torch.no_grad():
        xs[1].mul_(2.0)
    y2 = T.linear(xs[0], xs[1], xs[2], relu=relu)
    gx2 = torch.autograd.grad(y2, xs[0], dy.cuda())[0]
    mask = (y2.detach().cpu() > 0).double() if relu else torch.ones_like(yr)
    assert _rel(gx2, (dy.double() * mask) @ (2 * w.double())) < 5e-5


def test_packed_in_projection_slices():
    """The packed in_proj of nn.MultiheadAttention as row slices of one weight applied to different inputs (encoder: q|k from
    src + pos, v from src; decoder: three inputs with different row counts); the gradient of the weight is assembled in place."""
    g = _g(5)
    w, b = torch.randn(768, 256, generator=g) / 16, torch.randn(768, generator=g)
    xa, xb, xc = torch.randn(300, 256, generator=g), torch.randn(1024, 256, generator=g), torch.randn(1024, 256, generator=g)
    da, db_, dc = torch.randn(300, 256, generator=g), torch.randn(1024, 256, generator=g), torch.randn(1024, 256, generator=g)
    xs = [_leaf(w), _leaf(b), _leaf(xa), _leaf(xb), _leaf(xc)]
    ya, yb, yc = T.Proj.apply(xs[0], xs[1], ((0, 256), (256, 512), (512, 768)), False, 0.0, xs[2], xs[3], xs[4])
    grads = torch.autograd.grad([ya, yb, yc], xs, [da.cuda(), db_.cuda(), dc.cuda()])
    rs = [_leaf64(w), _leaf64(b), _leaf64(xa), _leaf64(xb), _leaf64(xc)]
    outs = [F.linear(rs[2], rs[0][:256], rs[1][:256]), F.linear(rs[3], rs[0][256:512], rs[1][256:512]),
            F.linear(rs[4], rs[0][512:], rs[1][512:])]
    refs = torch.autograd.grad(outs, rs, [da.double(), db_.double(), dc.double()])
    for got, want in zip([ya, yb, yc], outs):
        assert _rel(got, want) < 2e-5
    for got, want in zip(grads, refs):
        assert _rel(got, want) < 5e-5
    pos = torch.randn(512, 256, generator=g)
    xr = _leaf(xb)
    yr = T.AddRows.apply(xr, pos.cuda(), 512)
    assert _rel(yr, (xb.view(2, 512, 256) + pos).view(1024, 256)) < 1e-7
    assert torch.equal(torch.autograd.grad(yr, xr, db_.cuda())[0], db_.cuda())


@pytest.mark.parametrize('form', [0, 1, 2, 3])
@pytest.mark.parametrize('nb,nq,packed', [(2, 512, True), (2, 100, False), (1, 24, False), (3, 33, False), (2, 200, False), (24, 256, False), (16, 200, False),
                                          (12, 64, False), (7, 40, False)])
def test_attention_forward_backward(nb, nq, packed, form):
    """attn_train_fwd / attn_bwd_dq / attn_bwd_dkv (recompute-softmax backward) vs softmax attention under torch autograd; both
    forms of the kernels (knob train_attention_form; 0 = the shipped choice: the one-pass backward from 24 pairs x 256 queries up, the
    two-kernel second form below)."""
    from cotr_amd import _lib
    _lib.set_knob('train_attention_form', form)
    g = _g(nb * 1000 + nq)
    scale = 32 ** -0.5
    q = torch.randn(nb * nq, 256, generator=g) * 2
    k, v = torch.randn(nb * 512, 256, generator=g), torch.randn(nb * 512, 256, generator=g)
    d_o = torch.randn(nb * nq, 256, generator=g)
    rq, rk, rv = _leaf64(q), _leaf64(k), _leaf64(v)
    qh = rq.view(nb, nq, 8, 32).permute(0, 2, 1, 3) * scale
    kh = rk.view(nb, 512, 8, 32).permute(0, 2, 1, 3)
    vh = rv.view(nb, 512, 8, 32).permute(0, 2, 1, 3)
    o_ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).permute(0, 2, 1, 3).reshape(nb * nq, 256)
    gq, gk, gv = torch.autograd.grad(o_ref, [rq, rk, rv], d_o.double())
    if packed:
        qk = _leaf(torch.cat([q, k], dim=1))
        vv = _leaf(v)
        o = T.Attention.apply(qk, None, None, vv, nb, nq, scale, 0.0)
        dqk, dv = torch.autograd.grad(o, [qk, vv], d_o.cuda())
        dq, dk = dqk[:, :256], dqk[:, 256:]
    else:
        xs = [_leaf(q), _leaf(k), _leaf(v)]
        o = T.Attention.apply(None, xs[0], xs[1], xs[2], nb, nq, scale, 0.0)
        dq, dk, dv = torch.autograd.grad(o, xs, d_o.cuda())
    assert _rel(o, o_ref) < 2e-5
    assert _rel(dq, gq) < 5e-5 and _rel(dk, gk) < 5e-5 and _rel(dv, gv) < 5e-5


@pytest.mark.parametrize('nb,nq,packed', [(2, 512, True), (3, 200, False), (1, 33, False), (16, 200, False), (8, 100, False)])
def test_attention_backward_forms_agree_with_dropout(nb, nq, packed):
    """Dropout on the probabilities (p = 0.1): the two forms of the backward kernels regenerate the same mask from (seed, element
    index) and give the same dq / dk / dv up to summation order (the dropout path has no closed-form torch reference: the mask is
    this library's own counter-based one; checked statistically below and by finite differences in test_training_gpu.py)."""
    from cotr_amd import _lib
    g = _g(nb * 77 + nq)
    q, k, v = torch.randn(nb * nq, 256, generator=g), torch.randn(nb * 512, 256, generator=g), torch.randn(nb * 512, 256, generator=g)
    d_o = torch.randn(nb * nq, 256, generator=g).cuda()
    res = []
    for form in (1, 2, 3, 0):     # (3: one workgroup per (pair, head) when dq is packed, keys split over 2 / 4 otherwise; 0: the shipped choice)
        _lib.set_knob('train_attention_form', form)
        T.reseed(99)
        if packed:
            qk, vv = _leaf(torch.cat([q, k], dim=1)), _leaf(v)
            o = T.Attention.apply(qk, None, None, vv, nb, nq, 32 ** -0.5, 0.1)
            res.append((o.detach(),) + torch.autograd.grad(o, [qk, vv], d_o))
        else:
            xs = [_leaf(q), _leaf(k), _leaf(v)]
            o = T.Attention.apply(None, xs[0], xs[1], xs[2], nb, nq, 32 ** -0.5, 0.1)
            res.append((o.detach(),) + torch.autograd.grad(o, xs, d_o))
    for other in res[1:]:
        for a, b in zip(res[0], other):                          # output, then the gradients
            assert _rel(b, a) < 2e-5


def test_kv_block_used_by_two_attention_calls_sums_the_gradients():
    """ColBlocks hands every K / V column block an in-place gradient destination; it is claimed by the FIRST Attention that uses
    the block.  A second consumer of the same block (decode_train called twice with one kv list) must ADD its dk / dv, not
    overwrite the first one's: the gradient of the wide projection equals the one computed with separate, copied blocks."""
    nb, nq = 2, 40
    g = _g(77)
    scale = 32 ** -0.5
    wide = torch.randn(nb * 512, 1024, generator=g)                  # two layers' K | V columns
    q1, q2 = torch.randn(nb * nq, 256, generator=g), torch.randn(nb * nq, 256, generator=g)
    d1, d2 = torch.randn(nb * nq, 256, generator=g).cuda(), torch.randn(nb * nq, 256, generator=g).cuda()

    def run(shared):
        x = _leaf(wide)
        qa, qb = _leaf(q1), _leaf(q2)
        if shared:
            blocks = T.col_blocks(x, 1, 4)[0]                        # K0 V0 K1 V1 as views with gradient destinations
            k0, v0 = blocks[0], blocks[1]
            o1 = T.Attention.apply(None, qa, k0, v0, nb, nq, scale, 0.0)
            o2 = T.Attention.apply(None, qb, k0, v0, nb, nq, scale, 0.0)      # the SAME block objects a second time
        else:
            o1 = T.Attention.apply(None, qa, x[:, 0:256].contiguous(), x[:, 256:512].contiguous(), nb, nq, scale, 0.0)
            o2 = T.Attention.apply(None, qb, x[:, 0:256].contiguous(), x[:, 256:512].contiguous(), nb, nq, scale, 0.0)
        loss = (o1 * d1).sum() + (o2 * d2).sum()
        return torch.autograd.grad(loss, [x, qa, qb])
    gs, gp = run(True), run(False)
    assert float(gp[0][:, :512].abs().max()) > 0 and float(gs[0][:, 512:].abs().max()) == 0     # unused blocks: zero
    for a, b in zip(gs, gp):
        assert _rel(a, b) < 1e-5


def test_attention_dropout_statistics_and_determinism():
    """With dropout the kernel drops ~p of the probabilities and rescales by 1/(1-p): the output is an unbiased estimate of the
    un-dropped one; the same seed gives the same bits."""
    nb, nq = 2, 256
    g = _g(9)
    q, k = torch.randn(nb * nq, 256, generator=g).cuda(), torch.randn(nb * 512, 256, generator=g).cuda()
    v = torch.randn(nb * 512, 256, generator=g).cuda()
    base = T.Attention.apply(None, q, k, v, nb, nq, 0.05, 0.0)
    T.reseed(7)
    a = T.Attention.apply(None, q, k, v, nb, nq, 0.05, 0.1)
    T.reseed(7)
    b = T.Attention.apply(None, q, k, v, nb, nq, 0.05, 0.1)
    assert torch.equal(a, b) and not torch.equal(a, base)
    acc = torch.zeros_like(base)
    n = 40
    for i in range(n):
        acc += T.Attention.apply(None, q, k, v, nb, nq, 0.05, 0.1)
    # near-uniform attention over 512 keys with zero-mean values: one draw is off by sqrt(p / (1-p)) = 33 % of |o| (relative),
    # the mean of 40 draws by ~5 %; unbiased: the SIGNED mean deviation over the 131072 outputs is far smaller
    dev = acc / n - base
    assert 0.02 < float(dev.abs().mean() / base.abs().mean()) < 0.08
    assert abs(float(dev.mean())) < 0.01 * float(base.abs().mean())


@pytest.mark.parametrize('nb,nq', [(2, 100), (1, 1000), (3, 7)])
def test_output_head_forward_backward(nb, nq):
    g = _g(nb + nq)
    R = nb * nq
    h, w2, b2 = torch.randn(R, 256, generator=g), torch.randn(2, 256, generator=g) / 16, torch.randn(2, generator=g)
    dy = torch.randn(nb, nq, 2, generator=g)
    xs = [_leaf(h), _leaf(w2), _leaf(b2)]
    y = T.Head.apply(xs[0], xs[1], xs[2], nb, nq)
    grads = torch.autograd.grad(y, xs, dy.cuda())
    rs = [_leaf64(h), _leaf64(w2), _leaf64(b2)]
    yr = F.linear(*rs).view(nb, nq, 2)
    refs = torch.autograd.grad(yr, rs, dy.double())
    assert _rel(y, yr) < 2e-5
    for got, want in zip(grads, refs):
        assert _rel(got, want) < 3e-5


def test_relu_dropout_backward_and_transpose():
    from cotr_amd import _lib
    lib = _lib.load_library()
    g = _g(11)
    x = torch.randn(777, 1024, generator=g).cuda()
    w, b = (torch.randn(1024, 256, generator=g) / 16).cuda(), torch.zeros(1024).cuda()
    xin = torch.randn(777, 256, generator=g).cuda().requires_grad_()
    T.reseed(3)
    y = T.linear(xin, w, b, relu=True, p=0.25)
    frac = (y > 0).float().mean().item()
    assert 0.3 < frac < 0.45                                      # ~half positive, a quarter of those dropped
    dy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, xin, dy)
    want = ((y > 0).float() * dy / 0.75) @ w                      # the kept, positive elements pass dy / (1 - p)
    assert _rel(gx, want) < 5e-5
    t = torch.empty(1024, 777, device='cuda')
    assert lib.cotr_train_transpose(x.data_ptr(), t.data_ptr(), 777, 1024, _lib.current_stream_ptr()) == 0
    assert torch.equal(t, x.t().contiguous())


@pytest.mark.parametrize('B,H,cin,cout,k,stride,res,relu', [(2, 16, 128, 128, 3, 1, False, True), (1, 32, 256, 128, 3, 2, False, True),
                                                            (2, 16, 256, 512, 1, 2, False, False), (1, 16, 128, 512, 1, 1, True, True),
                                                            (3, 8, 512, 256, 1, 1, False, True),
                                                            # shapes whose weight gradient takes the implicit form (no im2col image, round 6)
                                                            (2, 32, 128, 128, 3, 1, True, True), (2, 64, 128, 128, 3, 2, False, True),
                                                            (4, 16, 256, 256, 3, 1, False, True), (4, 32, 512, 1024, 1, 2, False, False)])
def test_conv_frozenbn_forward_backward(B, H, cin, cout, k, stride, res, relu):
    """train_ops.ConvBN (layer2 / layer3 of the trainable backbone): forward = the inference conv kernel, backward = im2col +
    transpose-free wgrad + GEMM / col2im dgrad on the NHWC side-by-side layout; against F.conv2d on each 256-wide half (NCHW,
    fp64) with the FrozenBN affine, residual and ReLU of the bottleneck."""
    g = _g(B * 100 + H + cin + cout + k + stride)
    x = torch.randn(B, cin, H, 2 * H, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    sc, bi = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(B, cout, Ho, 2 * Ho, generator=g) if res else None
    dy = torch.randn(B, cout, Ho, 2 * Ho, generator=g)

    def nhwc(t):
        return t.permute(0, 2, 3, 1).contiguous()
    xs = [_leaf(nhwc(x)), _leaf(w)] + ([_leaf(nhwc(r))] if res else [])
    y = T.ConvBN.apply(xs[0], xs[1], sc.cuda(), bi.cuda(), xs[2] if res else None, relu, stride)
    grads = torch.autograd.grad(y, xs, nhwc(dy).cuda())
    rs = [_leaf64(x), _leaf64(w)] + ([_leaf64(r)] if res else [])
    halves = [F.conv2d(rs[0][..., :H], rs[1], stride=stride, padding=k // 2), F.conv2d(rs[0][..., H:], rs[1], stride=stride, padding=k // 2)]
    yr = torch.cat(halves, dim=-1) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
    if res:
        yr = yr + rs[2]
    if relu:
        yr = yr * (y.detach().permute(0, 3, 1, 2).cpu() > 0).double()      # the kernel's own mask (see test_linear_forward_backward)
    refs = torch.autograd.grad(yr, rs, dy.double())
    assert _rel(y, nhwc(yr.detach())) < 3e-5
    assert _rel(grads[0], nhwc(refs[0])) < 5e-5
    assert _rel(grads[1], refs[1]) < 5e-5
    if res:
        assert _rel(grads[2], nhwc(refs[2])) < 5e-5


@pytest.mark.parametrize('B,H,cin,cout,k,stride', [(2, 32, 128, 128, 3, 1), (2, 64, 128, 128, 3, 2), (4, 16, 256, 256, 3, 1),
                                                    (4, 32, 256, 256, 3, 2), (4, 32, 512, 1024, 1, 2), (16, 16, 256, 256, 3, 1),
                                                    (3, 32, 128, 128, 3, 1)])
def test_implicit_conv_wgrad_is_the_explicit_one_bit_for_bit(B, H, cin, cout, k, stride):
    """cotr_train_conv_wgrad_parts (round 6): the split-M partials of a convolution's weight gradient with the im2col image gathered by the
    kernel's own LDS-DMA (zeros where a tap leaves the 256-wide half) - the same operands in the same order as cotr_train_im2col +
    cotr_train_gemm_tn_parts: the same number of partials, the same bits; and the gradient they sum to against F.conv2d's in fp64."""
    from cotr_amd import _lib
    lib = _lib.load_library()
    sp = _lib.current_stream_ptr()
    g = _g(B + H + cin + cout + k + stride)
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, 2 * H, cin, generator=g).cuda()
    dz = torch.randn(B, Ho, 2 * Ho, cout, generator=g).cuda()
    m, kk = B * Ho * 2 * Ho, k * k * cin
    dz2 = dz.view(m, cout)
    got = T.conv_wgrad_parts(dz2, x, B, H, H, cin, cout, k, stride)
    assert got is not None, 'a shape of the trainable backbone at which the implicit form must apply'
    part, nparts, pstride = got
    col = torch.empty(m, kk, device='cuda')
    assert lib.cotr_train_im2col(x.data_ptr(), col.data_ptr(), B, H, H, cin, k, stride, sp) == 0
    part2, nparts2, pstride2 = T.gemm_tn_parts(dz2, col)
    assert (nparts, pstride) == (nparts2, pstride2)
    assert torch.equal(part[:nparts * pstride], part2[:nparts * pstride])
    dw = T.sum_parts(part, nparts, cout * kk).view(cout, k, k, cin).permute(0, 3, 1, 2)
    xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(False)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    dzr = dz.permute(0, 3, 1, 2).double().cpu()
    y = torch.cat([F.conv2d(xr[..., :H], w, stride=stride, padding=pad), F.conv2d(xr[..., H:], w, stride=stride, padding=pad)], dim=-1)
    (ref,) = torch.autograd.grad(y, w, dzr)
    assert _rel(dw, ref) < 5e-5
    # where the form does not apply the caller is told so (and forms the image): 64 input channels, a small product
    assert T.conv_wgrad_parts(torch.zeros(2 * 16 * 32, 64, device='cuda'), torch.zeros(2, 16, 32, 64, device='cuda'), 2, 16, 16, 64, 64, 3, 1) is None
    assert T.conv_wgrad_parts(torch.zeros(1 * 8 * 16, 256, device='cuda'), torch.zeros(1, 8, 16, 256, device='cuda'), 1, 8, 8, 256, 256, 3, 1) is None


@pytest.mark.parametrize('B,H,cin,planes,stride,down', [(2, 32, 512, 128, 1, False), (2, 64, 256, 128, 2, True), (4, 16, 1024, 256, 1, False),
                                                         (2, 32, 512, 256, 2, True)])
def test_bottleneck_as_one_autograd_node_is_the_four_convbn_nodes_bit_for_bit(B, H, cin, planes, stride, down):
    """train_ops.Bottleneck (round 6): a trainable ResNet bottleneck as ONE autograd node - the identity branch's gradient enters conv1's
    data-gradient GEMM as its residual instead of meeting conv1's gradient in an add launch of autograd's.  The same launches
    otherwise and one fp32 addition either way: output, input gradient and all weight gradients equal those of the four-ConvBN
    composition bit for bit (layer2 / layer3 shapes, with and without the downsample branch)."""
    g = _g(B + H + cin + planes + stride)
    cout = 4 * planes

    def wgt(o, i, k):
        return (torch.randn(o, i, k, k, generator=g) / math.sqrt(i * k * k)).cuda()

    def aff(c):
        return (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    ws = [wgt(planes, cin, 1), wgt(planes, planes, 3), wgt(cout, planes, 1)] + ([wgt(cout, cin, 1)] if down else [])
    affs = [aff(planes), aff(planes), aff(cout)] + ([aff(cout)] if down else [])
    x0 = torch.randn(B, H, 2 * H, cin, generator=g).cuda()
    Ho = H // stride
    dy = torch.randn(B, Ho, 2 * Ho, cout, generator=g).cuda()

    def run(one_node):
        x = x0.clone().requires_grad_()
        w = [t.clone().requires_grad_() for t in ws]
        if one_node:
            ds = (w[3], *affs[3]) if down else (None, None, None)
            y = T.Bottleneck.apply(x, w[0], *affs[0], w[1], *affs[1], w[2], *affs[2], *ds, stride)
        else:
            o = T.ConvBN.apply(x, w[0], *affs[0], None, True, 1)
            o = T.ConvBN.apply(o, w[1], *affs[1], None, True, stride)
            idt = T.ConvBN.apply(x, w[3], *affs[3], None, False, stride) if down else x
            y = T.ConvBN.apply(o, w[2], *affs[2], idt, True, 1)
        return [y.detach()] + list(torch.autograd.grad(y, [x] + w, dy))
    a, b = run(True), run(False)
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_reduce_jobs_kernel():
    """cotr_train_reduce_jobs against torch: several jobs in one launch - plain (vector path), a tail that is not a multiple of
    the chunk, two sources accumulating into one destination that already holds a value, misaligned records (scalar path: the
    head's 514-float partials), and a conv weight gradient (row scale + [row][tap][cin] -> [row][cin][tap])."""
    from cotr_amd import train_ops as T
    g = torch.Generator().manual_seed(5)
    dev = 'cuda'
    params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in ((256, 256), (256,), (2, 256), (2,), (64, 32, 3, 3), (3000,))]
    sink = T.GradSink(params)
    sink.flat.copy_(torch.randn(sink.flat.numel(), generator=g).to(dev))           # gradients start non-zero: the launch ACCUMULATES
    start = [p.grad.detach().clone() for p in params]
    want = [s.clone() for s in start]
    # Linear: dW | db in one partial record, used twice (9 and 3 partials)
    for nparts in (9, 3):
        part = torch.randn(nparts, 256 * 256 + 256, generator=g).to(dev)
        sink.add(params[0].grad, part, 0, nparts, 256 * 256 + 256, 256 * 256)
        sink.add(params[1].grad, part, 256 * 256, nparts, 256 * 256 + 256, 256)
        acc = torch.zeros(256 * 256 + 256, device=dev)
        for p in range(nparts):
            acc = acc + part[p]
        want[0] = want[0] + acc[:256 * 256].view(256, 256)
        want[1] = want[1] + acc[256 * 256:]
    # head: 514-float records (8-byte aligned only)
    part = torch.randn(5, 514, generator=g).to(dev)
    sink.add(params[2].grad, part, 0, 5, 514, 512)
    sink.add(params[3].grad, part, 512, 5, 514, 2)
    acc = torch.zeros(514, device=dev)
    for p in range(5):
        acc = acc + part[p]
    want[2] = want[2] + acc[:512].view(2, 256)
    want[3] = want[3] + acc[512:]
    # conv 3x3: packed [Cout][tap][Cin] partials, scaled per row, into torch's [Cout][Cin][3][3]
    scale = torch.rand(64, generator=g).to(dev) + 0.5
    part = torch.randn(17, 64 * 9 * 32, generator=g).to(dev)
    sink.add(params[4].grad, part, 0, 17, 64 * 9 * 32, 64 * 9 * 32, scale=scale, row_len=9 * 32, cin=32, taps=9)
    acc = torch.zeros(64 * 9 * 32, device=dev)
    for p in range(17):
        acc = acc + part[p]
    want[4] = want[4] + (acc.view(64, 9 * 32) * scale[:, None]).view(64, 9, 32).permute(0, 2, 1).reshape(64, 32, 3, 3)
    # a tail: 3000 elements = 2 full chunks + 952
    part = torch.randn(2, 3000, generator=g).to(dev)
    sink.add(params[5].grad, part, 0, 2, 3000, 3000)
    want[5] = want[5] + (part[0] + part[1])
    jobs, srcs, nchunks = sink.tables()
    assert len(jobs) == 6 and len(srcs) == 8 and nchunks == 64 + 1 + 1 + 1 + 18 + 3
    by_dst = {int(j['dst']): j for j in jobs}                                       # (the jobs are ordered longest walk first)
    assert [int(by_dst[p.grad.data_ptr()]['vec']) for p in params] == [1, 1, 0, 0, 1, 1]
    assert [int(by_dst[p.grad.data_ptr()]['n_src']) for p in params] == [2, 2, 1, 1, 1, 1]
    assert int(jobs['chunk0'][0]) == 0 and sum(int(n) for n in srcs['nparts'][:int(jobs['n_src'][0])]) == 17
    cmap = T.GradSink.chunk_map(jobs, nchunks)
    assert len(cmap) == nchunks and all(jobs['chunk0'][cmap[c]] <= c for c in range(nchunks)) and cmap[-1] == len(jobs) - 1
    untouched = sink.flat.clone()
    sink.flush()
    torch.cuda.synchronize()
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w), (tuple(p.shape), float((p.grad - w).abs().max()))
    # nothing outside the six gradients was written (the padding between them)
    mask = torch.ones_like(untouched, dtype=torch.bool)
    for p in params:
        o = (p.grad.data_ptr() - sink.flat.data_ptr()) // 4
        mask[o:o + p.numel()] = False
    assert torch.equal(sink.flat[mask], untouched[mask])
    sink.flush()                                                                    # nothing registered: no launch, no change
    assert sink.last == (0, 0)


def test_derived_weight_operands_one_launch_refresh():
    """train_ops._derived: the W^T slices of the dX GEMMs, the packed [Cout][k][k][Cin] convolution weights and the FrozenBN-scaled
    transposed ones are kept in persistent buffers and re-derived from the parameters by ONE launch (cotr_train_perm_jobs) after the
    optimiser step.  Values = what the single-purpose kernels / torch produce, bit for bit (a transpose is a copy; the scaling is
    one fp32 multiply); an in-place parameter update is picked up by refresh_derived() and the buffers stay where they are."""
    g = _g(123)
    T.clear_weight_cache()
    w = torch.nn.Parameter(torch.randn(768, 256, generator=g).cuda())
    wc3 = torch.nn.Parameter(torch.randn(128, 64, 3, 3, generator=g).cuda())
    wc1 = torch.nn.Parameter(torch.randn(96, 160, 1, 1, generator=g).cuda())
    sc3, sc1 = (torch.rand(128, generator=g) + 0.5).cuda(), (torch.rand(96, generator=g) + 0.5).cuda()

    def want():
        out = {'t_all': w.detach().t().contiguous(), 't_qk': w.detach()[0:512].t().contiguous(), 't_v': w.detach()[512:768].t().contiguous()}
        out['p3'] = wc3.detach().permute(0, 2, 3, 1).reshape(128, 576).contiguous()
        out['s3'] = (out['p3'] * sc3[:, None]).t().contiguous()
        out['s1'] = (wc1.detach().reshape(96, 160) * sc1[:, None]).t().contiguous()
        return out

    def got():
        return {'t_all': T.weight_t(w), 't_qk': T.weight_t(w[0:512]), 't_v': T.weight_t(w[512:768]), 'p3': T.conv_packed(wc3),
                's3': T.conv_scaled_t(wc3, sc3), 's1': T.conv_scaled_t(wc1, sc1)}
    a, b = got(), want()
    for k in b:
        assert torch.equal(a[k], b[k]), k
    assert T.conv_packed(wc1).data_ptr() == wc1.data_ptr()              # 1x1: the parameter is its own packed form
    ptrs = {k: v.data_ptr() for k, v in a.items()}
    with torch.no_grad():                                                # what FusedAdam does: the weights change through raw pointers
        w.data.add_(1.0)
        wc3.data.mul_(0.5)
        wc1.data.sub_(0.25)
    w._version, wc3._version                                             # (versions moved here; a captured step would not move them)
    assert T.refresh_derived() == 6
    a, b = got(), want()
    for k in b:
        assert torch.equal(a[k], b[k]), k
        assert a[k].data_ptr() == ptrs[k], k                             # persistent buffers: a captured graph can keep reading them
    T.mark_derived_stale()
    with torch.no_grad():
        w.data.add_(1.0)
    assert torch.equal(T.weight_t(w[0:512]), w.detach()[0:512].t().contiguous())      # a stale entry is re-derived at its next use
    T.clear_weight_cache()
