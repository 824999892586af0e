import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
import cotr_amd
from cotr_amd import training
from cotr_amd.models import build_model
from tests.golden.make_train_golden import train_case
g = np.load('tests/golden/train_step_b2_q24.npz')
sd, img, query, target = train_case()
for fwd in (training.forward_train, training.forward_train_torch):
    m = build_model(cotr_amd.default_args(dropout=0.0)).cuda(); m.load_state_dict(sd); m.train()
    i, q = img.cuda(), query.cuda()
    feats = training.backbone_features(m, i)
    pred = fwd(m, i, q, feats); cycle = fwd(m, i, pred, feats)
    mask = torch.norm(cycle - q, dim=-1) < 10 / 256
    cl = F.mse_loss(cycle[mask], q[mask])
    named = dict(m.named_parameters()); names = [str(n) for n in g['cgrad_names']]
    grads = torch.autograd.grad(cl, [named[n] for n in names])
    rel = [(abs(float(gr.double().norm()) - n) / n, nm) for nm, gr, (s, a, n) in zip(names, grads, g['cgrad_stats'])]
    rel.sort(reverse=True)
    print(fwd.__name__, 'cycle_loss', cl.item(), float(g['cycle_loss']), 'worst:', [(round(r, 5), nm) for r, nm in rel[:12]], 'median', np.median([r for r, _ in rel]))
