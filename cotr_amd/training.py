"""Training step: the reference's stage 1 (frozen backbone, ``train_cotr.py --lr_backbone=0``) and its
stages 2-3 (``--lr_backbone > 0``: layer2 / layer3 of the backbone train, backbone.py:64-69) on hand-written HIP kernels,
forward AND backward.

What runs where:
* backbone (ResNet-50 to layer3, FrozenBN; 70 % of the forward FLOPs, no gradient in stage 1): the inference kernels,
  through ``cotr_backbone`` (C ABI) - once per step, shared by the prediction and the cycle pass;
* the whole trainable part - input_proj, 6 encoder layers, 6 decoder layers, decoder.norm, corr_embed - forward and
  backward: HIP kernels under a torch autograd tape (``cotr_amd/train_ops.py``): fp32-MFMA GEMMs for every contraction
  (forward, dX on the cached W^T, dW = dY^T . X by a transpose-free split-M kernel), attention with dropout and its
  recompute-softmax backward (dQ / dK,dV kernels), residual + dropout + LayerNorm fused forward and backward, ReLU + dropout,
  bias gradients, the lin_sine encodings and the 256 -> 2 head.  Between ``forward_train``'s entry and the loss PyTorch
  allocates tensors and records the tape; it computes nothing.  The loss (two ``mse_loss`` and a mask, cotr_trainer.py:
  124-135) and Adam stay torch;
* stages 2-3 (``--lr_backbone > 0``): conv1 + layer1 (frozen, backbone.py:66-69) on the inference kernels, layer2 / layer3
  under autograd on the HIP kernels as well (``train_ops.Bottleneck``, one autograd node per block: implicit-GEMM forward; dW by the
  transpose-free TN kernel that gathers its im2col operand itself; dX of the 3x3 stride-1 convolutions as one implicit-GEMM launch
  on the flipped kernel, of the others as a GEMM on the cached W^T [+ col2im]); the torch-convolution form is kept as a cross-check only
  (``backbone_features_trainable(..., use_torch_convs=True)``).
``forward_train_torch`` is the round-1 tape (HIP GEMMs + torch ops for everything else); it is kept as an independent
cross-check of the kernels (tests) and is not used by the product path.

Semantics follow the reference line by line: ``COTR.forward`` (COTR/models/cotr_model.py:26-40) with dropout active
(COTR/models/transformer.py:143-159,185-201; ``nn.MultiheadAttention(dropout=...)``), ``COTRTrainer.train_batch``
(COTR/trainers/cotr_trainer.py:118-150), the checkpoint dictionary of ``save_model`` (:75-88).  Because
``model(img, queries)`` works in training mode and the parameters are ordinary ``nn.Parameter`` objects, the reference's own
trainer / ``torch.optim.Adam(optim_list)`` (train_cotr.py:49-57) also run unchanged on this model.  Dropout masks come from a
counter-based generator of our own (not torch's stream): same distribution, different draws.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib

MAX_SIZE = 256
TOK = 512          # 16 x 32 feature positions of a side-by-side pair
CFEAT = 1024


# ----------------------------------------------------------------------------------------------------------------------
def _gemm(a, w, bias=None):
    """y[M,N] = a[M,K] . w[N,K]^T (+ bias) on the library's GEMM kernels; fp32, contiguous CUDA tensors."""
    lib = _lib.load_library()
    m, k = a.shape
    n = w.shape[0]
    assert w.shape[1] == k and k % 32 == 0 and n % 16 == 0, (a.shape, w.shape)
    y = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if m == 0:
        return y
    with torch.cuda.device(a.device):
        rc = lib.cotr_op_linear(a.data_ptr(), None, 0, w.data_ptr(), None, None if bias is None else bias.data_ptr(), None, 0,
                                y.data_ptr(), m, n, k, _lib.current_stream_ptr())
    if rc != 0:
        raise _lib.CotrHipError(f'cotr_op_linear failed (code {rc}) for M,N,K = {m},{n},{k}')
    return y


def _t_pad(t):
    """[M,C] -> contiguous [C, M rounded up to 32] (zero padded): the K-major operand of dW = dY^T . X."""
    m, c = t.shape
    mp = (m + 31) // 32 * 32
    out = torch.zeros((c, mp), dtype=t.dtype, device=t.device)
    out[:, :m] = t.t()
    return out


class _HipLinear(torch.autograd.Function):
    """nn.Linear on the HIP GEMM kernels, forward and backward."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return _gemm(x, w, None if b is None else b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _gemm(dy, w.t().contiguous())            # [M,N] . (W^T)[K,N]^T
        if ctx.needs_input_grad[1]:
            dw = _gemm(_t_pad(dy), _t_pad(x))             # (dY^T)[N,M] . (X^T)[K,M]^T
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def hip_linear(x, w, b=None):
    return _HipLinear.apply(x, w, b)


# ----------------------------------------------------------------------------------------------------------------------
def lin_sine(x, depth=64):
    """NerfPositionalEncoding 'lin_sine' (COTR/models/position_encoding.py:30-45): cat([sin(k pi x)]_k, [cos(k pi x)]_k)
    over a last dimension of size 2, k = 1..depth.  The reference evaluates it under ``torch.no_grad()``; callers here
    do the same (``forward_train``)."""
    kpi = torch.tensor([(i + 1) * math.pi for i in range(depth)], dtype=x.dtype, device=x.device)
    arg = x.unsqueeze(-2) * kpi.view(-1, 1)                       # [..., depth, 2]
    return torch.cat([torch.sin(arg).flatten(-2), torch.cos(arg).flatten(-2)], dim=-1)


def image_pos_table(device, h=16, w=32, hidden=256):
    """PositionEmbeddingSine with an all-False mask (position_encoding.py:60-72), fp32 like the reference -> [h*w, hidden]."""
    ones = torch.ones(1, h, w, dtype=torch.bool, device=device)
    y = ones.cumsum(1, dtype=torch.float32)
    x = ones.cumsum(2, dtype=torch.float32)
    y = (y - 0.5) / (y[:, -1:, :] + 1e-6)
    x = (x - 0.5) / (x[:, :, -1:] + 1e-6)
    return lin_sine(torch.stack([x, y], dim=-1), hidden // 4)[0].reshape(h * w, hidden)


def _heads(t, b, length, nheads):
    return t.view(b, length, nheads, -1).permute(0, 2, 1, 3)        # [B, heads, L, hd]


def _attention(q, k, v, b, lq, lk, nheads, p_drop, training):
    """softmax(q k^T) v per head; q already scaled (nn.MultiheadAttention scales q by head_dim^-0.5 before q.k^T)."""
    qh, kh, vh = _heads(q, b, lq, nheads), _heads(k, b, lk, nheads), _heads(v, b, lk, nheads)
    attn = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1)
    attn = F.dropout(attn, p_drop, training)
    return (attn @ vh).permute(0, 2, 1, 3).reshape(b * lq, -1)


def _ln(x, norm):
    return F.layer_norm(x, (x.shape[-1],), norm.weight, norm.bias, norm.eps)


def _ffn(x, layer, p_drop, training):
    hid = F.dropout(F.relu(hip_linear(x, layer.linear1.weight, layer.linear1.bias)), p_drop, training)
    return hip_linear(hid, layer.linear2.weight, layer.linear2.bias)


def backbone_features(model, img):
    """layer3 features [B*512, 1024] of the frozen backbone on the HIP kernels (no gradient)."""
    lib = model._ensure_ready(img.device)
    img = img.detach().contiguous().float()
    feat = torch.empty((img.shape[0] * TOK, CFEAT), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        model._ensure_workspace(lib, img.device, img.shape[0], 1)
        _lib.check(lib.cotr_backbone(model._handle, img.data_ptr(), img.shape[0], feat.data_ptr(), _lib.current_stream_ptr()),
                   model._handle, 'cotr_backbone')
    return feat


def _frozen_bn(x, bn):
    """FrozenBatchNorm2d.forward (COTR/models/backbone.py:46-56): buffers only, no gradient of its own."""
    scale = bn.weight * (bn.running_var + 1e-5).rsqrt()
    return x * scale.view(1, -1, 1, 1) + (bn.bias - bn.running_mean * scale).view(1, -1, 1, 1)


def _bottleneck(x, blk):
    """torchvision ResNet v1.5 Bottleneck (stride on the 3x3) with FrozenBN, as the reference's backbone body builds it."""
    out = F.relu(_frozen_bn(blk.conv1(x), blk.bn1))
    out = F.relu(_frozen_bn(blk.conv2(out), blk.bn2))
    out = _frozen_bn(blk.conv3(out), blk.bn3)
    idt = _frozen_bn(blk.downsample[0](x), blk.downsample[1]) if hasattr(blk, 'downsample') else x
    return F.relu(out + idt)


def _frozen_bn_affine(bn):
    """(scale, bias) of a FrozenBatchNorm2d (backbone.py:46-56): buffers only -> constants, computed once per module and kept
    ON the module (a process-wide dict keyed by id(bn) would hand a new module that reuses a dead one's address - with the same
    version counters - the dead one's affine, and keep its device tensors alive).  Re-derived when a buffer is replaced
    (.to() / load_state_dict assign new tensors or bump versions) or moved."""
    bufs = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version, t.device) for t in bufs)
    hit = bn.__dict__.get('_hip_affine')
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale = bn.weight * (bn.running_var + 1e-5).rsqrt()
            hit = (key, scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous())
        bn.__dict__['_hip_affine'] = hit          # plain attribute: not a buffer, not in the state dict
    return hit[1], hit[2]


def _bottleneck_hip(x, blk):
    """torchvision ResNet v1.5 Bottleneck (stride on the 3x3) with FrozenBN on the NHWC side-by-side layout, every convolution
    forward and backward on the HIP kernels (train_ops.Bottleneck: the block as one autograd node; train_ops.ConvBN: one node per convolution)."""
    from . import train_ops as T
    stride = blk.conv2.stride[0]
    if T.BOTTLENECK_FN:   # the block as one autograd node (round 6): the identity gradient rides in conv1's data-gradient GEMM
        ds = (blk.downsample[0].weight, *_frozen_bn_affine(blk.downsample[1])) if hasattr(blk, 'downsample') else (None, None, None)
        return T.Bottleneck.apply(x, blk.conv1.weight, *_frozen_bn_affine(blk.bn1), blk.conv2.weight, *_frozen_bn_affine(blk.bn2),
                                  blk.conv3.weight, *_frozen_bn_affine(blk.bn3), *ds, stride)
    out = T.ConvBN.apply(x, blk.conv1.weight, *_frozen_bn_affine(blk.bn1), None, True, 1)
    out = T.ConvBN.apply(out, blk.conv2.weight, *_frozen_bn_affine(blk.bn2), None, True, stride)
    idt = x
    if hasattr(blk, 'downsample'):
        idt = T.ConvBN.apply(x, blk.downsample[0].weight, *_frozen_bn_affine(blk.downsample[1]), None, False, stride)
    return T.ConvBN.apply(out, blk.conv3.weight, *_frozen_bn_affine(blk.bn3), idt, True, 1)


def backbone_features_trainable(model, img, use_torch_convs=False):
    """Stages 2-3 of the reference's recipe (--lr_backbone > 0): only layer2 / layer3 train (backbone.py:66-69), so conv1 +
    layer1 run on the inference kernels (cotr_backbone_upto, no gradient) and layer2 / layer3 run under autograd - on the HIP
    kernels forward and backward (train_ops.ConvBN, NHWC side-by-side: both halves of a pair in one tensor, each padded on its
    own).  ``use_torch_convs``: the round-1 path (torch / MIOpen convolutions per 256-wide half), kept as a cross-check.
    -> [B*512, 1024] with graph."""
    lib = model._ensure_ready(img.device)
    img = img.detach().contiguous().float()
    b = img.shape[0]
    l1 = torch.empty((b, 64, 128, 256), dtype=torch.float32, device=img.device)             # NHWC over the pair
    with torch.cuda.device(img.device):
        model._ensure_workspace(lib, img.device, b, 1)
        _lib.check(lib.cotr_backbone_upto(model._handle, img.data_ptr(), b, 1, l1.data_ptr(), _lib.current_stream_ptr()),
                   model._handle, 'cotr_backbone_upto')
    body = model.backbone[0].body
    if not use_torch_convs:
        y = l1
        for blk in list(body.layer2) + list(body.layer3):
            y = _bottleneck_hip(y, blk)
        return y.view(b * TOK, CFEAT)                                                        # [B,16,32,1024] = the token matrix
    x = l1.permute(0, 3, 1, 2)                                                               # [B,256,64,128]
    halves = []
    for half in (x[..., :64], x[..., 64:]):
        y = half
        for blk in list(body.layer2) + list(body.layer3):
            y = _bottleneck(y, blk)
        halves.append(y)
    feat = torch.cat(halves, dim=-1)                                                         # [B,1024,16,32]
    return feat.permute(0, 2, 3, 1).reshape(b * TOK, CFEAT)


def _backbone_trains(model):
    return any(p.requires_grad for p in model.backbone.parameters())


_pos_tables = {}


def _pos_table(device, d):
    """The image position table [512, d] of this device: a constant, built once (reference: recomputed every call under no_grad)."""
    key = (str(device), d)
    if key not in _pos_tables:
        with torch.no_grad():
            _pos_tables[key] = image_pos_table(device, hidden=d).contiguous()
    return _pos_tables[key]


def _query_encoding(queries):
    """lin_sine encoding of the queries on the HIP kernel (cotr_model.py:34-36); no gradient, like the reference
    (position_encoding.py:40-45)."""
    lib = _lib.load_library()
    pts = queries.detach().reshape(-1, 2).float().contiguous()
    out = torch.empty((pts.shape[0], 256), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = lib.cotr_op_posenc(pts.data_ptr(), out.data_ptr(), pts.shape[0], _lib.current_stream_ptr())
    if rc != 0:
        raise _lib.CotrHipError(f'cotr_op_posenc failed (code {rc})')
    return out


def encode_train(model, features, b):
    """input_proj + the 6 encoder layers (cotr_model.py:37, transformer.py:143-159) on ``features`` [b*512, 1024] ->
    (memory, memory + pos), each [b*512, 256], with an autograd graph; every operation is a HIP kernel (train_ops.py)."""
    from . import train_ops as T
    tr = model.transformer
    nheads, d = tr.nhead, tr.d_model
    assert d == 256 and nheads == 8
    scale = float(d // nheads) ** -0.5
    pos = _pos_table(features.device, d)
    src = T.linear(features, model.input_proj.weight.view(d, CFEAT), model.input_proj.bias)            # cotr_model.py:37
    for layer in tr.encoder.layers:                                                                    # transformer.py:143-159
        p = float(layer.self_attn.dropout) if model.training else 0.0
        w, bias = layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias
        qk, v = T.Proj.apply(w, bias, ((0, 2 * d), (2 * d, 3 * d)), False, 0.0, T.AddRows.apply(src, pos, TOK), src)
        ao = T.Attention.apply(qk, None, None, v, b, TOK, scale, p)
        ao = T.linear(ao, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
        src = T.AddDropLN.apply(src, ao, layer.norm1.weight, layer.norm1.bias, p)
        hid = T.linear(src, layer.linear1.weight, layer.linear1.bias, relu=True, p=p)
        src = T.AddDropLN.apply(src, T.linear(hid, layer.linear2.weight, layer.linear2.bias), layer.norm2.weight, layer.norm2.bias, p)
    return src, T.AddRows.apply(src, pos, TOK)


HOIST_KV = True     # decoder k / v projections of all layers (and both passes) as two GEMMs (train_ops.ProjKV); False: per layer


def decoder_kv_train(model, memory, mem_pos, passes=1):
    """k / v of every decoder layer for all rows of ``memory`` [passes * B * 512, 256] (the rows of pass h are block h) from two
    GEMMs (train_ops.ProjKV) -> list over passes of lists over layers of (k, v), each a [B*512, 256] column block (a view with
    leading dimension 6*256 that the attention kernels read in place)."""
    from . import train_ops as T
    layers = model.transformer.decoder.layers
    wb = []
    for layer in layers:
        wb += [layer.multihead_attn.in_proj_weight, layer.multihead_attn.in_proj_bias]
    k_all, v_all = T.ProjKV.apply(mem_pos, memory, *wb)
    kb, vb = T.col_blocks(k_all, passes, len(layers)), T.col_blocks(v_all, passes, len(layers))
    return [[(kb[h][l], vb[h][l]) for l in range(len(layers))] for h in range(passes)]


def decode_train(model, memory, mem_pos, queries, kv=None):
    """The 6 cross-attention decoder layers, decoder.norm and corr_embed (transformer.py:185-201,110-111, cotr_model.py:34-39)
    for ``queries`` [B,Q,2] against ``memory`` [B*512, 256] -> pred_corrs [B,Q,2].  The query encoding carries no gradient
    (``NerfPositionalEncoding.forward`` is ``@torch.no_grad()``, COTR/models/position_encoding.py:40-45): in the cycle pass
    ``model(img, pred)`` nothing flows back through ``pred``.  ``kv``: this pass's entry of ``decoder_kv_train`` (then ``memory`` /
    ``mem_pos`` are not read here)."""
    from . import train_ops as T
    tr = model.transformer
    nheads, d = tr.nhead, tr.d_model
    scale = float(d // nheads) ** -0.5
    b, nq, _ = queries.shape
    query_pos = _query_encoding(queries)                                                               # cotr_model.py:34-36
    if kv is None and HOIST_KV:
        kv = decoder_kv_train(model, memory, mem_pos, 1)[0]
    tgt = None                                                                                         # zeros, transformer.py:54
    for li, layer in enumerate(tr.decoder.layers):                                                     # transformer.py:185-201
        p = float(layer.multihead_attn.dropout) if model.training else 0.0
        w, bias = layer.multihead_attn.in_proj_weight, layer.multihead_attn.in_proj_bias
        xq = query_pos if tgt is None else T.AddRows.apply(tgt, query_pos, 0)
        if kv is not None:
            (q,), (k, v) = T.Proj.apply(w, bias, ((0, d),), False, 0.0, xq), kv[li]
        else:
            q, k, v = T.Proj.apply(w, bias, ((0, d), (d, 2 * d), (2 * d, 3 * d)), False, 0.0, xq, mem_pos, memory)
        ao = T.Attention.apply(None, q, k, v, b, nq, scale, p)
        ao = T.linear(ao, layer.multihead_attn.out_proj.weight, layer.multihead_attn.out_proj.bias)
        tgt = T.AddDropLN.apply(tgt, ao, layer.norm2.weight, layer.norm2.bias, p)
        hid = T.linear(tgt, layer.linear1.weight, layer.linear1.bias, relu=True, p=p)
        tgt = T.AddDropLN.apply(tgt, T.linear(hid, layer.linear2.weight, layer.linear2.bias), layer.norm3.weight, layer.norm3.bias, p)
    hs = T.AddDropLN.apply(None, tgt, tr.decoder.norm.weight, tr.decoder.norm.bias, 0.0)   # only the last layer reaches the loss
    mlp = model.corr_embed.layers                                                          # position_encoding.py:23-26
    x = T.linear(hs, mlp[0].weight, mlp[0].bias, relu=True)
    x = T.linear(x, mlp[1].weight, mlp[1].bias, relu=True)
    return T.Head.apply(x, mlp[2].weight, mlp[2].bias, b, nq)


def forward_train(model, img, queries, features=None):
    """``COTR.forward`` in training mode -> pred_corrs [B,Q,2] with an autograd graph over the trainable part; every
    operation of the graph is a HIP kernel (cotr_amd/train_ops.py).  Row layout is batch-major: row b*L + l is token / query
    l of pair b."""
    if features is None:
        features = backbone_features_trainable(model, img) if _backbone_trains(model) else backbone_features(model, img)
    memory, mem_pos = encode_train(model, features, queries.shape[0])
    return decode_train(model, memory, mem_pos, queries)


def forward_train_torch(model, img, queries, features=None, _query_grad=False):
    """Round-1 tape, kept as an independent cross-check (tests): HIP GEMMs, everything else torch ops.  Not the product path.
    Row layout is batch-major: row b*L + l is token / query l of pair b.

    The query encoding carries NO gradient: ``NerfPositionalEncoding.forward`` is ``@torch.no_grad()``
    (COTR/models/position_encoding.py:40-45), so in the cycle pass ``model(img, pred)`` of the trainer
    (cotr_trainer.py:129) nothing flows back through ``pred`` into the first pass - the cycle term only trains the
    second pass.  ``_query_grad=True`` re-enables that (non-reference) path; it exists so that the test which pins
    this behaviour can show that the golden tells the two apart."""
    tr = model.transformer
    nheads, d = tr.nhead, tr.d_model
    scale = float(d // nheads) ** -0.5
    training = model.training
    b, nq, _ = queries.shape
    if features is None:
        features = backbone_features_trainable(model, img) if _backbone_trains(model) else backbone_features(model, img)
    pos = image_pos_table(img.device, hidden=d)                                                        # [512, d]
    src = hip_linear(features, model.input_proj.weight.view(d, CFEAT), model.input_proj.bias)          # cotr_model.py:37

    def add_pos(x):
        return (x.view(b, TOK, d) + pos).view(b * TOK, d)

    for layer in tr.encoder.layers:                                                                    # transformer.py:143-159
        p = layer.self_attn.dropout
        w, bias = layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias
        qk = hip_linear(add_pos(src), w[:2 * d], bias[:2 * d])
        v = hip_linear(src, w[2 * d:], bias[2 * d:])
        ao = _attention(qk[:, :d] * scale, qk[:, d:], v, b, TOK, TOK, nheads, p, training)
        ao = hip_linear(ao, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
        src = _ln(src + F.dropout(ao, p, training), layer.norm1)
        src = _ln(src + F.dropout(_ffn(src, layer, p, training), p, training), layer.norm2)
    memory, mem_pos = src, add_pos(src)

    if _query_grad:
        query_pos = lin_sine(queries.reshape(-1, 2).float(), d // 4)
    else:
        with torch.no_grad():                                                                          # position_encoding.py:40
            query_pos = lin_sine(queries.detach().reshape(-1, 2).float(), d // 4)                      # cotr_model.py:34-36
    tgt = torch.zeros_like(query_pos)                                                                  # transformer.py:54
    for layer in tr.decoder.layers:                                                                    # transformer.py:185-201
        p = layer.multihead_attn.dropout
        w, bias = layer.multihead_attn.in_proj_weight, layer.multihead_attn.in_proj_bias
        q = hip_linear(tgt + query_pos, w[:d], bias[:d]) * scale
        k = hip_linear(mem_pos, w[d:2 * d], bias[d:2 * d])
        v = hip_linear(memory, w[2 * d:], bias[2 * d:])
        ao = _attention(q, k, v, b, nq, TOK, nheads, p, training)
        ao = hip_linear(ao, layer.multihead_attn.out_proj.weight, layer.multihead_attn.out_proj.bias)
        tgt = _ln(tgt + F.dropout(ao, p, training), layer.norm2)
        tgt = _ln(tgt + F.dropout(_ffn(tgt, layer, p, training), p, training), layer.norm3)
    hs = _ln(tgt, tr.decoder.norm)                       # only the last layer's output reaches the loss (cotr_model.py:39)
    mlp = model.corr_embed.layers                        # position_encoding.py:23-26
    x = F.relu(hip_linear(hs, mlp[0].weight, mlp[0].bias))
    x = F.relu(hip_linear(x, mlp[1].weight, mlp[1].bias))
    return F.linear(x, mlp[2].weight, mlp[2].bias).view(b, nq, 2)


# ----------------------------------------------------------------------------------------------------------------------
def compute_loss(model, img, query, target, cycle_consis=True, bidirectional=True, branch_free=False):
    """The loss of ``COTRTrainer.train_batch`` / ``validate_batch`` (cotr_trainer.py:124-142) -> (loss, pred).
    ``branch_free`` writes the masked cycle term without the reference's ``if mask.sum() > 0`` / boolean indexing (a host
    synchronisation and a data-dependent shape, neither of which a captured HIP graph can hold): the masked squared error
    summed and divided by the number of selected elements, 0 when none is selected - the same value."""
    # no dropout / batch statistics in the backbone: the prediction and the cycle pass see the same features
    feat_fn = backbone_features_trainable if _backbone_trains(model) else backbone_features
    feats = feat_fn(model, img)
    b = img.shape[0]
    if not cycle_consis:
        pred = forward_train(model, img, query, feats)
        return F.mse_loss(pred, target), pred
    # The cycle pass decodes the FIRST pass's prediction, but its encoder does not depend on it: both passes' encoders run as
    # ONE batch of 2B pairs (each half with its own dropout realisation, as two separate calls would have) - half the encoder
    # launches, GEMMs of twice the rows, one gradient accumulation per encoder parameter instead of two.
    if bidirectional:
        feats2 = feats                                                    # same image: cycle = model(img, pred)
    else:
        img_rev = torch.cat([img[..., MAX_SIZE:], img[..., :MAX_SIZE]], dim=-1)
        feats2 = feat_fn(model, img_rev)
    memory, mem_pos = encode_train(model, torch.cat([feats, feats2], dim=0), 2 * b)
    rows = b * TOK
    kv = decoder_kv_train(model, memory, mem_pos, 2) if HOIST_KV else (None, None)      # neither pass's k / v depends on its queries
    pred = decode_train(model, memory[:rows], mem_pos[:rows], query, kv[0])
    loss = F.mse_loss(pred, target)
    if bidirectional:
        cycle = decode_train(model, memory[rows:], mem_pos[rows:], pred, kv[1])
    else:
        q_rev = pred.clone()
        q_rev[..., 0] = q_rev[..., 0] - 0.5
        cycle = decode_train(model, memory[rows:], mem_pos[rows:], q_rev, kv[1])
        cycle = torch.stack([cycle[..., 0] - 0.5, cycle[..., 1]], dim=-1)
    mask = torch.norm(cycle - query, dim=-1) < 10 / MAX_SIZE
    if branch_free:
        # select BEFORE squaring: a NaN / inf in an entry the mask rejects must not reach the sum (NaN * 0 is NaN) nor its
        # gradient - cycle[mask] in the reference excludes such entries from both (its norm compares False: rejected)
        diff = torch.where(mask.unsqueeze(-1), cycle - query, torch.zeros_like(cycle))
        loss = loss + (diff ** 2).sum() / (2.0 * mask.sum()).clamp(min=1.0)
    elif mask.sum() > 0:
        loss = loss + F.mse_loss(cycle[mask], query[mask])
    return loss, pred


class GraphedTrainStep:
    """One optimisation step - zero_grad, ``compute_loss``, backward, [gradient averaging], ``optim.step()`` - captured ONCE as a
    HIP graph and replayed: the ~1000 launches of a step cost one graph launch of host time instead of ~20 ms of Python.

        step = GraphedTrainStep(model, optimizer_for(model, capturable=True), img, query, target)
        loss, pred = step(img, query, target)            # tensors; loss.item() when the value is needed

    What capture needs and how it is met: static input buffers (copied into per call); no allocation or synchronisation
    inside the library (workspace from torch's allocator, cotr_set_workspace); the loss without host branches
    (``branch_free``); Adam with ``capturable=True``; and dropout masks that change from replay to replay although every
    launch's seed argument is frozen in the graph - the training kernels XOR their seed with a salt word in device memory
    (``cotr_train_set_dropout_salt``) that the captured step advances itself.
    Difference to ``train_batch``: the reference skips backward when the loss is NaN (cotr_trainer.py:145-147); a graph cannot
    skip, so ``__call__(..., check=True)`` raises after the fact instead (the weights have then seen the NaN step)."""

    def __init__(self, model, optim, img, query, target, cycle_consis=True, bidirectional=True, group=None, warmup=3, sink=True):
        from . import train_ops as T
        assert model.training and img.is_cuda
        assert all(g.get('capturable', False) for g in optim.param_groups), 'build the optimiser with capturable=True'
        self.model, self.optim, self.group = model, optim, group
        # the step's own GradSink (its job table is baked into the graph: it is never used for eager steps)
        self.sink = grad_sink_for(optim) if sink else None
        self.args = (cycle_consis, bidirectional)
        self.img, self.query, self.target = img.clone(), query.clone(), target.clone()
        self.salt = torch.zeros(1, dtype=torch.int32, device=img.device)
        lib = _lib.load_library()
        # the captured graph bakes in the addresses of the library's workspace: it must not be replaced (a larger eval batch
        # between replays) while this object is alive - pin it; a call that would have to grow it raises instead
        model.pin_workspace(self)
        _lib.check(lib.cotr_train_set_dropout_salt(self.salt.data_ptr()), None, 'cotr_train_set_dropout_salt')
        side = torch.cuda.Stream(device=img.device)
        side.wait_stream(torch.cuda.current_stream(img.device))
        with torch.cuda.stream(side):                      # PyTorch's whole-network capture recipe: warm up on a side stream
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(img.device).wait_stream(side)
        torch.cuda.synchronize(img.device)
        # (the derived weight operands - train_ops._derived - were made by the warm-up steps and are persistent: the captured step
        #  reads them and, with a FusedAdam, re-derives them itself behind its optimiser step; any other optimiser: see __call__)
        self.graph = torch.cuda.CUDAGraph()
        if self.sink is None:
            optim.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss, self.pred = self._body()
        if self.sink is not None:
            self.sink.frozen = True                        # the graph replays its table upload from the sink's pinned buffer
        self._derived_keys = T.derived_keys()              # what a replay keeps current by itself (it re-derives them behind its update)
        # ... and the graph holds the ADDRESSES of their buffers and of the cached job table: keep them alive for as long as it can
        # be replayed, whatever a later miss, refresh or clear_weight_cache() does to the registry (released in close())
        self._derived_hold = T.hold_derived()

    def _body(self):
        self.salt.add_(0x3C6EF35F)                         # a new mask family per step (int32 wrap-around is fine)
        if self.sink is not None:
            self.sink.attach()
            self.sink.zero()
        else:
            self.optim.zero_grad(set_to_none=True)
        loss, pred = compute_loss(self.model, self.img, self.query, self.target, *self.args, branch_free=True)
        if self.sink is not None:
            with self.sink.collect():
                loss.backward()
        else:
            loss.backward()
        import torch.distributed as dist
        if (self.group is not None or (dist.is_available() and dist.is_initialized())) and self.sink is not None:
            from .dist import sync_flat_gradients
            sync_flat_gradients(self.sink.flat, self.group)
        elif self.group is not None or (dist.is_available() and dist.is_initialized()):
            sync_gradients([p for g in self.optim.param_groups for p in g['params']], self.group)
        self.optim.step()
        if not isinstance(self.optim, FusedAdam):           # (a FusedAdam re-derives the weight operands behind its own update)
            from . import train_ops as T
            T.refresh_derived()
        return loss.detach(), pred.detach()

    def __call__(self, img, query, target, check=False):
        assert self.graph is not None, 'GraphedTrainStep was closed'
        self.img.copy_(img)
        self.query.copy_(query)
        self.target.copy_(target)
        self.graph.replay()
        # a replay updates the weights without bumping their Python version counters: derived operands the captured step does not
        # refresh itself (made by an eager step in between, or all of them with an optimiser other than FusedAdam) are stale now
        from . import train_ops as T
        T.mark_derived_stale(keep=self._derived_keys)
        if check and not bool(torch.isfinite(self.loss)):
            raise FloatingPointError('loss is not finite in a captured training step (train_batch would have skipped it)')
        return self.loss, self.pred

    def close(self):
        """Detach the salt word from the training kernels (eager steps afterwards use their seeds alone).  Also run when the
        object is collected: the kernels must not keep reading a word whose tensor is gone."""
        if getattr(self, 'salt', None) is not None:
            try:
                with torch.cuda.device(self.salt.device):     # only if the registered word is still ours
                    _lib.load_library().cotr_train_clear_dropout_salt(self.salt.data_ptr())
            except Exception:
                pass
            self.salt = None
            self.graph = None          # (the captured step writes the salt word: it must not be replayed after this)
            self._derived_hold = None  # the graph is gone: the operand buffers / job table it addressed may go too
            try:
                self.model.unpin_workspace(self)
            except Exception:
                pass

    def __del__(self):
        self.close()


def grad_sink_for(optim):
    """A ``train_ops.GradSink`` over the optimiser's parameters: their gradients become views of one flat buffer and a backward
    pass finishes all of them with one reduction launch (``train_batch(..., sink=...)``).  A ``FusedAdam`` owns one already."""
    from . import train_ops as T
    own = getattr(optim, 'sink', None)
    if own is not None:
        return own
    return T.GradSink([p for g in optim.param_groups for p in g['params']])


class FusedAdam(torch.optim.Adam):
    """``torch.optim.Adam`` (train_cotr.py:49-57) whose step is ONE kernel launch (``cotr_train_adam``) over every trainable
    parameter: the gradients are the flat buffer of a ``GradSink`` (``self.sink`` - pass it to ``train_batch``), exp_avg /
    exp_avg_sq are flat buffers of the same layout.  Same update as torch's (multi-tensor) Adam up to the rounding of fused
    multiply-adds (tests/test_training_gpu.py: 1e-6 of the weights' scale over ten steps); torch's own step is ~15 multi-tensor launches that each stream
    the state (0.57 ms per stage-1 step, 0.73 ms per stage-2 step).  ``state_dict()`` / ``load_state_dict()`` have torch's layout
    (per parameter ``step``, ``exp_avg``, ``exp_avg_sq``), so optimiser checkpoints are interchangeable with the reference's.
    ``zero_grad()`` zeroes the flat buffer (the gradients stay views of it); a step with a gradient that is not the sink's view
    raises."""

    def __init__(self, groups, capturable=False, **kw):
        from . import train_ops as T
        super().__init__(groups, capturable=capturable, foreach=None, **kw)
        for g in self.param_groups:
            assert not g['amsgrad'] and g['weight_decay'] == 0 and not g['maximize'], 'FusedAdam covers the recipe of train_cotr.py only'
        assert len(self.param_groups) <= 8
        self.sink = T.GradSink([p for g in self.param_groups for p in g['params']])
        flat = self.sink.flat
        self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        self._capturable = bool(capturable)
        # ONE step count for all parameters, where torch keeps it: on the device for capturable steps, on the host otherwise
        self._step = torch.zeros((), dtype=torch.float32, device=flat.device if capturable else 'cpu')
        self._adopt_state()
        # job table: one record per parameter, one workgroup per 1024 elements
        group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g['params']}
        dt = np.dtype([('p', '<u8'), ('off', '<u8'), ('numel', '<u4'), ('chunk0', '<u4'), ('group', '<u4'), ('vec', '<u4')])
        jobs = np.zeros(len(self.sink.params), dtype=dt)
        chunk = 0
        for i, (p, off) in enumerate(zip(self.sink.params, self.sink.offsets)):
            jobs[i] = (p.data_ptr(), off, p.numel(), chunk, group_of[id(p)], int(p.data_ptr() % 16 == 0))
            chunk += (p.numel() + 1023) // 1024
        cmap = T.GradSink.chunk_map(jobs, chunk)
        self._nchunks = chunk
        self._ptrs = [p.data_ptr() for p in self.sink.params]
        self._jobs = torch.from_numpy(jobs.view(np.uint8).copy()).to(flat.device)
        self._cmap = torch.from_numpy(cmap.view(np.uint8).copy()).to(flat.device)

    def _adopt_state(self, loaded=None):
        """state[p] = torch's entries as views of the flat buffers (``loaded``: a state just read by load_state_dict - copied in)."""
        steps = set()
        for p, off in zip(self.sink.params, self.sink.offsets):
            m = self._m[off:off + p.numel()].view_as(p)
            v = self._v[off:off + p.numel()].view_as(p)
            st = None if loaded is None else loaded.get(p)
            if st:
                m.copy_(st['exp_avg'])
                v.copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
            elif loaded is not None:
                m.zero_()
                v.zero_()
            self.state[p] = {'step': self._step, 'exp_avg': m, 'exp_avg_sq': v}
        if loaded is not None:
            assert len(steps) <= 1, 'FusedAdam keeps one step count for all parameters'
            self._step.fill_(float(steps.pop() if steps else 0))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        loaded = {p: dict(self.state[p]) for p in self.sink.params if p in self.state and 'exp_avg' in self.state[p]}
        self._adopt_state(loaded)

    def zero_grad(self, set_to_none=True):
        self.sink.attach()
        self.sink.zero()

    def _check(self):
        lo = self.sink.flat.data_ptr()
        for p, off, ptr in zip(self.sink.params, self.sink.offsets, self._ptrs):
            if p.grad is None or p.grad.data_ptr() != lo + off * 4:
                raise RuntimeError('FusedAdam: a gradient is not the view of the flat buffer (use optim.zero_grad() / sink.zero(), '
                                   'not p.grad = None)')
            if p.data_ptr() != ptr:
                raise RuntimeError('FusedAdam: a parameter was re-allocated after the optimiser was built')

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self._check()
        from . import train_ops as T
        lib = _lib.load_library()
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        assert all(g['betas'] == g0['betas'] and g['eps'] == g0['eps'] for g in self.param_groups)
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g['lr']) for g in self.param_groups])
        self._step += 1
        if self._capturable:                    # the host does not know the step count of a replayed graph
            bc1, bc2s, step_ptr = 1.0, 1.0, ctypes.c_void_p(self._step.data_ptr())
        else:
            t = int(self._step.item())          # (a host tensor: no synchronisation)
            bc1, bc2s, step_ptr = 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5, None
        with T._on(self.sink.flat.device):
            T._chk(lib.cotr_train_adam(ctypes.c_void_p(self._jobs.data_ptr()), ctypes.c_void_p(self._cmap.data_ptr()), self._nchunks,
                                       ctypes.c_void_p(self.sink.flat.data_ptr()), ctypes.c_void_p(self._m.data_ptr()),
                                       ctypes.c_void_p(self._v.data_ptr()), lrs, len(self.param_groups), float(b1), float(b2),
                                       float(g0['eps']), bc1, bc2s, step_ptr, T._sp()), 'cotr_train_adam')
        # the kernel wrote the weights through raw pointers (their Python version counters did not move): every weight-shaped operand
        # derived from them - W^T slices, packed / BN-scaled convolution weights - is re-derived now, in one launch
        T.refresh_derived()
        return None


def train_batch(model, optim, img, query, target, cycle_consis=True, bidirectional=True, group=None, sink=None, defer_check=True):
    """One optimisation step, ``COTRTrainer.train_batch`` (cotr_trainer.py:118-150); with ``group`` (or an initialised
    default process group) the gradients are averaged over the ranks (one rank per GPU, RCCL) before the optimiser
    step.  -> (loss value, pred).

    The reference reads the loss back (``loss.data.item()``, :144) and skips backward + lets the optimiser step on zeroed gradients
    when it is NaN (:145-147).  ``defer_check`` (default) keeps that OUTCOME - a NaN step leaves the weights alone, every other step is
    the reference's - but not the host's waiting: backward is enqueued right behind the forward, the loss is read back while it runs, and
    a NaN step's gradients are thrown away afterwards; the cycle term is written without the reference's ``if mask.sum() > 0`` (a second
    read-back in the middle of the step; ``compute_loss(branch_free=True)``: the same value).  The GPU then never waits for Python between
    forward and backward (16 pairs x 200 queries: 18.1 -> 16.9 ms per stage-1 step).  ``defer_check=False``: the reference's control flow
    literally (read back, then decide whether to run backward).

    Rank symmetry: with several ranks the skip decision has to be COMMON - a rank that skipped would leave the others alone in the
    gradient collective - so the NaN flag is max-reduced first and every rank skips (or steps) together.

    ``sink`` (``grad_sink_for(optim)``, made once and passed to every step; a ``FusedAdam`` brings its own): the gradients live in the
    sink's flat buffer, are zeroed by one memset and finished by one reduction launch after backward instead of a reduction + an
    accumulation per weight - same values bit for bit.  A NaN step then leaves the weights alone (no optimiser step), as
    ``zero_grad()`` to None does on the path without a sink."""
    import torch.distributed as dist
    assert model.training
    distributed = group is not None or (dist.is_available() and dist.is_initialized())
    if sink is None:
        sink = getattr(optim, 'sink', None)              # a FusedAdam brings its own
    if sink is not None:
        sink.attach()
        sink.zero()
    else:
        optim.zero_grad()

    # several ranks + a sink: the gradient exchange starts from inside the sink's flush - the buffer is reduced in 3 address-ordered
    # ranges, each range's reduce-scatter / all-gather issued right behind its reduction launch (dist.flat_exchange_async), so the
    # exchange overlaps the rest of the reduction instead of following it.  (With a deferred NaN check the exchange of a step that
    # turns out to be skipped has already run - on every rank alike, so the collectives stay matched - and is simply not used.)
    exchange_in_flush = distributed and sink is not None and dist.get_world_size(group) > 1

    def backward(loss):
        if sink is not None:
            if exchange_in_flush:
                from .dist import flat_exchange_async
                sink.set_exchange(flat_exchange_async(group), parts=3)
            try:
                with sink.collect():
                    loss.backward()
            finally:
                sink.set_exchange(None)
        else:
            loss.backward()

    def is_bad(loss):
        value = loss.item()
        bad = math.isnan(value)
        if distributed and dist.get_world_size(group) > 1:
            flag = torch.tensor([1.0 if bad else 0.0], device=loss.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            bad = bool(flag.item() > 0)
        return value, bad

    if defer_check:
        loss, pred = compute_loss(model, img, query, target, cycle_consis, bidirectional, True)
        backward(loss)                                   # enqueued behind the forward; the read-back below overlaps with it
        value, bad = is_bad(loss)
    else:
        loss, pred = compute_loss(model, img, query, target, cycle_consis, bidirectional)
        value, bad = is_bad(loss)
        if not bad:
            backward(loss)
    if bad:
        if sink is not None:
            return value, pred.detach()                  # (the sink's buffer is zeroed at the start of the next step)
        optim.zero_grad()
    elif distributed and sink is not None:
        if not exchange_in_flush:                        # (world size 1: the exchange is the identity; kept for the RCCL tests)
            from .dist import sync_flat_gradients
            sync_flat_gradients(sink.flat, group)
    elif distributed:
        sync_gradients([p for g in optim.param_groups for p in g['params']], group)
    optim.step()
    if not isinstance(optim, FusedAdam):                 # (a FusedAdam re-derives the weight operands itself, behind its update)
        from . import train_ops as T
        T.refresh_derived()
    return value, pred.detach()


def sync_gradients(params, group=None, bucket_elems=1 << 25):
    """Average the gradients over the ranks: reduce-scatter + all-gather of flat fp32 buckets (cotr_amd/dist.py)."""
    from .dist import sync_gradients_sharded
    sync_gradients_sharded(params, group, bucket_elems)


def optimizer_for(model, learning_rate=1e-4, lr_backbone=0.0, capturable=False, fused=False):
    """``torch.optim.Adam(optim_list)`` of train_cotr.py:49-57 - group for group, including the EMPTY ``query_proj``
    group (the encoding has no parameters), so that ``optim_state_dict`` of a reference checkpoint loads here and the
    other way round (Adam requires the same number of param groups)."""
    groups = [{'params': list(model.transformer.parameters()), 'lr': learning_rate},
              {'params': list(model.corr_embed.parameters()), 'lr': learning_rate},
              {'params': list(model.query_proj.parameters()), 'lr': learning_rate},
              {'params': list(model.input_proj.parameters()), 'lr': learning_rate}]
    if lr_backbone > 0:
        groups.append({'params': list(model.backbone.parameters()), 'lr': lr_backbone})
    if fused:       # same groups, same state dict; the step is one kernel launch on the GradSink's flat buffers (optim.sink)
        return FusedAdam(groups, capturable=capturable)
    return torch.optim.Adam(groups, capturable=capturable)


def save_checkpoint(path, model, optim, epoch, iteration):
    """The dictionary of ``COTRTrainer.save_model`` (cotr_trainer.py:75-88)."""
    torch.save({'epoch': epoch, 'iteration': iteration, 'optim_state_dict': optim.state_dict(),
                'model_state_dict': model.state_dict()}, path)


def load_checkpoint(path, model, optim=None):
    ck = torch.load(path, map_location='cpu')
    model.load_state_dict(ck['model_state_dict'])
    if optim is not None and 'optim_state_dict' in ck:
        optim.load_state_dict(ck['optim_state_dict'])
    return ck.get('epoch', 0), ck.get('iteration', 0)
