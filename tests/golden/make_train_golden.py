"""Golden loss and gradients of one training step computed by the REFERENCE model itself (COTR/models imported unchanged
behind the stubs of oracle/ref_import.py) on CPU, with the loss of COTRTrainer.train_batch (COTR/trainers/cotr_trainer.py:
124-135, cycle_consis and bidirectional on), dropout 0 so that the step is deterministic, backbone frozen (stage 1).
Authoring container only.      python tests/golden/make_train_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from cotr_amd.utils.synth import synth_state_dict  # noqa: E402

FULL = ['corr_embed.layers.2.weight', 'corr_embed.layers.2.bias', 'transformer.decoder.norm.weight', 'input_proj.bias',
        'transformer.encoder.layers.0.self_attn.in_proj_bias', 'transformer.decoder.layers.5.multihead_attn.out_proj.weight',
        'transformer.encoder.layers.5.linear1.bias']


def train_case(seed=0, B=2, Q=24):
    """Weights: the synthetic state with the last corr_embed layer damped so that every answer lands near c = (0.30, 0.60):
    queries near c then pass the cycle test (|f(f(q)) - q| < 10/256) and the cycle term is part of the loss."""
    sd = synth_state_dict(seed)
    sd['corr_embed.layers.2.weight'] = sd['corr_embed.layers.2.weight'] * 0.01
    sd['corr_embed.layers.2.bias'] = torch.tensor([0.30, 0.60])
    g = torch.Generator().manual_seed(seed + 50)
    img = torch.randn(B, 3, 256, 512, generator=g)
    query = torch.tensor([0.30, 0.60]) + 0.02 * (torch.rand(B, Q, 2, generator=g) - 0.5)
    query[:, ::5] = torch.rand(B, (Q + 4) // 5, 2, generator=g)           # some queries far away: masked out of the cycle term
    target = torch.rand(B, Q, 2, generator=g)
    return sd, img, query, target


def main(lr_backbone=0.0, out_name='train_step_b2_q24.npz'):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd, img, query, target = train_case()
    model = ref_import.build_reference_model(ref_import.default_args(dropout=0.0, lr_backbone=lr_backbone))
    model.load_state_dict(sd)
    model.train()
    # cotr_trainer.py:124-135
    pred = model(img, query)['pred_corrs']
    loss = torch.nn.functional.mse_loss(pred, target)
    cycle = model(img, pred)['pred_corrs']
    mask = torch.norm(cycle - query, dim=-1) < 10 / 256
    assert 0 < int(mask.sum()) < mask.numel(), int(mask.sum())
    cycle_loss = torch.nn.functional.mse_loss(cycle[mask], query[mask])
    # the cycle term alone (it is ~1e-4 of the loss; invisible in the total).  NerfPositionalEncoding.forward is
    # @torch.no_grad() (position_encoding.py:40-45): no gradient flows through the prediction fed back as queries, the
    # cycle term only trains the second pass.  With the last layer damped x0.01 the blocked path would be ~4e-4 of this
    # gradient anyway - train_step_cycle_b2_q24.npz (cycle_main below) is the case that can tell the two apart.
    params = [(n, p) for n, p in model.named_parameters()
              if p.requires_grad and not ('decoder' in n and 'norm1' in n) and 'layer4' not in n]
    cgrads = torch.autograd.grad(cycle_loss, [p for _, p in params], retain_graph=True)
    loss = loss + cycle_loss
    loss.backward()
    out = {'loss': np.array(loss.item()), 'cycle_loss': np.array(cycle_loss.item()), 'pred': pred.detach().numpy(),
           'mask': mask.numpy()}
    names, stats = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            assert not p.requires_grad or ('norm1' in name and 'decoder' in name) or 'layer4' in name, name
            continue
        gr = p.grad.double()
        names.append(name)
        stats.append([float(gr.sum()), float(gr.abs().sum()), float(gr.norm())])
        if name in FULL:
            out['grad.' + name] = p.grad.numpy()
    out['cgrad_names'] = np.array([n for n, _ in params])
    out['cgrad_stats'] = np.array([[float(g.double().sum()), float(g.double().abs().sum()), float(g.double().norm())] for g in cgrads])
    for (n, _), g in zip(params, cgrads):
        if n in FULL:
            out['cgrad.' + n] = g.numpy()
    out['grad_names'] = np.array(names)
    out['grad_stats'] = np.array(stats)
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print('loss', loss.item(), 'cycle', cycle_loss.item(), 'mask', int(mask.sum()), '/', mask.numel(), 'params with grad', len(names))


def cycle_case(model=None, seed=1, B=2, Q=24, gain=3.0):
    """A case where the path the reference BLOCKS (cycle loss -> pred -> query encoding -> first pass) would be a large
    part of the cycle gradient: nothing damped, q/k projections sharpened x3 so that the answer really depends on the
    query (|d pred / d query| ~ 0.7 instead of 0.04), queries placed around each pair's fixed point f(q*) = q* (found by
    iterating the reference model) so that they pass the cycle test.  Queries / targets are stored in the golden (they
    depend on the reference model's outputs)."""
    sd = synth_state_dict(seed, attn_gain=gain)
    sd['corr_embed.layers.2.bias'] = torch.tensor([0.35, 0.55])
    g = torch.Generator().manual_seed(seed + 70)
    img = torch.randn(B, 3, 256, 512, generator=g)
    if model is None:
        return sd, img
    model.load_state_dict(sd)
    with torch.no_grad():
        q = torch.tensor([0.35, 0.55]).repeat(B, 1, 1)
        for _ in range(30):
            q = model(img, q)['pred_corrs']
    query = q + 0.01 * (torch.rand(B, Q, 2, generator=g) - 0.5)
    query[:, ::6] = torch.rand(B, (Q + 5) // 6, 2, generator=g)
    target = torch.rand(B, Q, 2, generator=g)
    return sd, img, query, target


def cycle_main(out_name='train_step_cycle_b2_q24.npz'):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = ref_import.build_reference_model(ref_import.default_args(dropout=0.0, lr_backbone=0.0))
    model.train()
    sd, img, query, target = cycle_case(model)
    params = [(n, p) for n, p in model.named_parameters()
              if p.requires_grad and not ('decoder' in n and 'norm1' in n) and 'layer4' not in n]

    def cycle_terms():
        pred = model(img, query)['pred_corrs']
        cycle = model(img, pred)['pred_corrs']
        mask = torch.norm(cycle - query, dim=-1) < 10 / 256
        cl = torch.nn.functional.mse_loss(cycle[mask], query[mask])
        return pred, mask, cl, torch.autograd.grad(cl, [p for _, p in params], allow_unused=True)

    pred, mask, cycle_loss, cgrads = cycle_terms()                     # the reference as it is
    assert 0 < int(mask.sum()) < mask.numel(), int(mask.sum())
    loss = torch.nn.functional.mse_loss(pred, target) + cycle_loss
    # for comparison only: the same with the no_grad of the query encoding lifted (what round 1 of cotr_amd computed)
    enc = model.query_proj
    blocked = type(enc).forward
    type(enc).forward = blocked.__wrapped__
    try:
        _, mask2, cl2, cgrads2 = cycle_terms()
    finally:
        type(enc).forward = blocked
    assert torch.equal(mask, mask2) and abs(cl2.item() - cycle_loss.item()) < 1e-9

    def stats(gs):
        return np.array([[0.0, 0.0, 0.0] if g is None else [float(g.double().sum()), float(g.double().abs().sum()),
                                                           float(g.double().norm())] for g in gs])
    s1, s2 = stats(cgrads), stats(cgrads2)
    rel = np.abs(s2[:, 2] - s1[:, 2]) / np.maximum(s1[:, 2], 1e-30)
    out = {'loss': np.array(loss.item()), 'cycle_loss': np.array(cycle_loss.item()), 'pred': pred.detach().numpy(),
           'mask': mask.numpy(), 'query': query.numpy(), 'target': target.numpy(),
           'cgrad_names': np.array([n for n, _ in params]), 'cgrad_stats': s1, 'cgrad_stats_query_grad': s2}
    for (n, _), g in zip(params, cgrads):
        if n in FULL and g is not None:
            out['cgrad.' + n] = g.numpy()
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print('cycle case: loss', loss.item(), 'cycle', cycle_loss.item(), 'mask', int(mask.sum()), '/', mask.numel(),
          '| gradient-norm change if the query encoding were differentiable: median %.3f, min %.3f, max %.3f'
          % (np.median(rel), rel.min(), rel.max()))


if __name__ == '__main__':
    if '--cycle-only' in sys.argv:
        cycle_main()
        sys.exit(0)
    main()
    cycle_main()
    # stages 2-3 of the reference's recipe: layer2 / layer3 of the backbone train as well (backbone.py:66-69)
    FULL[:] = ['corr_embed.layers.2.bias', 'input_proj.bias', 'backbone.0.body.layer2.0.conv1.weight',
               'backbone.0.body.layer2.3.conv3.weight']
    main(lr_backbone=1e-5, out_name='train_step_backbone_b2_q24.npz')
