"""Interleaved in-process A/B of knob settings at one shape: ms per forward for the defaults and for every given setting, R rounds
(one model, one box, settings alternate inside a round so that clock / thermal drift hits all of them alike).  GPU box.

    python tools/ab_inproc.py B Q [--rounds 5] [--iters 200] [--check] SETTING [SETTING ...]
        SETTING = knob=value[,knob=value...]        e.g.  side_stream=1  side_stream=3  xcd_mapping=33

--check also compares every setting's output with the defaults' (max |difference| in pixels, and whether it is bit-identical).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

ap = argparse.ArgumentParser()
ap.add_argument('B', type=int)
ap.add_argument('Q', type=int)
ap.add_argument('settings', nargs='*')
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--iters', type=int, default=0)
ap.add_argument('--check', action='store_true')
a = ap.parse_intermixed_args()
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
img, qs = synth_inputs(a.B, a.Q, seed=1)
img, qs = img.cuda(), qs.cuda()
iters = a.iters or max(5, min(200, int(200 / a.B)))
variants = [('default', {})] + [(s, {kv.split('=')[0]: int(kv.split('=')[1]) for kv in s.split(',')}) for s in a.settings]


def apply(kn):
    m.reset_knobs()
    for k, v in kn.items():
        m.set_knob(k, v)


def run(n):
    for _ in range(3):
        m(img, qs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        m(img, qs)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


times = {name: [] for name, _ in variants}
for r in range(a.rounds):
    for name, kn in (variants if r % 2 == 0 else variants[::-1]):
        apply(kn)
        times[name].append(run(iters))
base = sorted(times['default'])[len(times['default']) // 2]
print(f'# B={a.B} Q={a.Q}, {a.rounds} rounds x {iters} forwards, median (min) ms per forward')
for name, _ in variants:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print(f'{name:48s} {med:9.4f} ({t[0]:9.4f})  {med / base - 1:+7.2%}')
if a.check:
    apply({})
    ref = m(img, qs)['pred_corrs'].clone()
    scale = torch.tensor([512.0, 256.0], device=ref.device)
    for name, kn in variants[1:]:
        apply(kn)
        outs = [m(img, qs)['pred_corrs'].clone() for _ in range(3)]
        err = float(((outs[0] - ref).abs() * scale).max())
        rep = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f'check {name:42s} max |d| vs default {err:.2e} px, bit-identical to default: {torch.equal(outs[0], ref)}, repeatable: {rep}')
m.reset_knobs()
