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
    # the cycle term alone (it is ~1e-4 of the loss: its gradient - through the prediction fed back as queries and the
    # derivative of the lin_sine encoding - would be invisible in the total)
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


if __name__ == '__main__':
    main()
    # stages 2-3 of the reference's recipe: layer2 / layer3 of the backbone train as well (backbone.py:66-69)
    FULL[:] = ['corr_embed.layers.2.bias', 'input_proj.bias', 'backbone.0.body.layer2.0.conv1.weight',
               'backbone.0.body.layer2.3.conv3.weight']
    main(lr_backbone=1e-5, out_name='train_step_backbone_b2_q24.npz')
