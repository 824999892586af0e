"""Decode time by pair count at Q queries per pair (cotr_decode alone, after one cotr_encode; knob batch_split off = one pass up to 32768
rows), what the shipped prefix rule of knob batch_split makes of it, and what a dynamic programme over the measured times would.  GPU box.
    python tools/decode_cost.py [Q=1000] [max pairs=32]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = build_model(cotr_amd.default_args()).cuda().eval()
m.load_state_dict(synth_state_dict(0))
img, qs = synth_inputs(NMAX, Q, seed=1)
img, qs = img.cuda(), qs.cuda()
m.reserve(NMAX, Q)


def time_decode(b, split):
    m.set_knob('batch_split', split)
    m.encode(img[:b])
    q = qs[:b].contiguous()
    for _ in range(3):
        m.decode(q)
    torch.cuda.synchronize()
    n = max(6, min(60, int(60 / b)))
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            m.decode(q)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


T = [0.0] + [time_decode(b, 0) for b in range(1, NMAX + 1)]
S = [0.0] + [time_decode(b, 1) for b in range(1, NMAX + 1)]
eff, first = [0.0] * (NMAX + 1), [0] * (NMAX + 1)
for n in range(1, NMAX + 1):
    best_c, best_t = n, T[n]
    for c in range(n - 1, 0, -1):
        t = T[c] + eff[n - c]
        if t < best_t - 1e-9:
            best_c, best_t = c, t
    if best_t > 0.98 * T[n]:
        best_c, best_t = n, T[n]
    first[n], eff[n] = best_c, best_t
print(f'# decode of b pairs x {Q} queries (ms): one pass | shipped batch_split (prefix rule) | dynamic programme over the one-pass times')
for n in range(1, NMAX + 1):
    parts, r = [], n
    while r:
        parts.append(first[r])
        r -= first[r]
    print(f'{n:3d} pairs: {T[n]:7.3f} | {S[n]:7.3f} ({100 * (S[n] / T[n] - 1):+5.1f} %) | {eff[n]:7.3f} ({100 * (eff[n] / T[n] - 1):+5.1f} %)  ' + ' + '.join(map(str, parts)))
