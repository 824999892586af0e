"""Measure the encode time of 1 ... 64 pairs (one pass each, knob batch_split off) and derive the table behind knob batch_split:
kEncFirst[n] = pairs of the FIRST pass of the cheapest partition of n pairs (csrc/enc_split.inc, api.hip enc_next_chunk).  The time
against the pair count is a staircase (tiles quantise to rounds of the 256 CUs): just above a step, the step + a small remainder is
cheaper than one pass.  A split is taken only where it wins at least 2 %.  GPU box.
    python tools/batch_cost.py [--write]      # --write: rewrite csrc/enc_split.inc (then rebuild: python -m cotr_amd.build)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

NMAX = 64
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
m.set_knob('batch_split', 0)
img, _ = synth_inputs(NMAX, 1, seed=1)
img = img.cuda()
m.reserve(NMAX, 1)


def time_encode(b):
    x = img[:b]
    for _ in range(3):
        m.encode(x)
    torch.cuda.synchronize()
    n = max(6, min(60, int(80 / b)))
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            m.encode(x)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


T = [0.0] + [time_encode(b) for b in range(1, NMAX + 1)]
eff, first = [0.0] * (NMAX + 1), [0] * (NMAX + 1)
for n in range(1, NMAX + 1):
    best_c, best_t = n, T[n]
    for c in range(n - 1, 0, -1):           # the large pass first; ties keep the larger first pass
        t = T[c] + eff[n - c]
        if t < best_t - 1e-9:
            best_c, best_t = c, t
    if best_t > 0.98 * T[n]:                # a split must win 2 %
        best_c, best_t = n, T[n]
    first[n], eff[n] = best_c, best_t
print('# encode time by pair count, one pass (ms), and the partition knob batch_split walks (ms, gain)')
for n in range(1, NMAX + 1):
    parts, r = [], n
    while r:
        parts.append(first[r])
        r -= first[r]
    print(f'{n:3d} pairs: {T[n]:8.3f} ms   ' + (f'-> {" + ".join(map(str, parts))}: {eff[n]:8.3f} ms  {100 * (eff[n] / T[n] - 1):+.1f} %' if len(parts) > 1 else ''))
if '--write' in sys.argv:
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cotr_amd', 'csrc', 'enc_split.inc')
    with open(path, 'w') as f:
        f.write('// kEncFirst[n], n = 1 ... 64: pairs of the FIRST encode pass of the cheapest partition of n pairs (api.hip enc_next_chunk), derived by\n'
                '// tools/batch_cost.py from the encode times it measured on an MI355X (one pass of n pairs each; a split must win 2 %):\n')
        f.write('//   ms per pass: ' + ' '.join(f'{t:.3f}' for t in T[1:]) + '\n')
        f.write('static const unsigned char kEncFirst[65] = {0, ' + ', '.join(str(c) for c in first[1:]) + '};\n')
    print('wrote', path)
