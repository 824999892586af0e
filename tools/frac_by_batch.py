"""Fraction of the fp32-MFMA peak along the batch axis: forward time at B pairs x Q queries for the default dispatch and for
alternative knob settings (which kernel family serves 1024 < rows < 8192 - FasterSparseEngine's grouped calls, every partial last
batch; sparse_engine.py:339-369, 400-411).  GPU box.

    python tools/frac_by_batch.py                       # the curve under the shipped defaults: B in 1..32 x Q in {1, 257, 1000}
    python tools/frac_by_batch.py --sweep               # ... and every candidate setting below next to it, the best one marked
    python tools/frac_by_batch.py --pairs 2,4,8 --queries 1000 --set ffn_fusion_max_rows=4096,attention_fusion_max_rows=4096
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cotr_amd
from cotr_amd.models import build_model
from cotr_amd.utils.synth import synth_state_dict, synth_inputs

BIG = 1 << 30
CANDIDATES = {
    # the small-row fused kernels (attention + out-proj partials / fused FFN + ln_reduce) pushed to more rows
    'fused<=2048': {'ffn_fusion_max_rows': 2048, 'attention_fusion_max_rows': 2048},
    'fused<=4096': {'ffn_fusion_max_rows': 4096, 'attention_fusion_max_rows': 4096},
    'ffn_fused<=4096': {'ffn_fusion_max_rows': 4096},
    'att_fused<=4096': {'attention_fusion_max_rows': 4096},
    'fused<=8192': {'ffn_fusion_max_rows': 8192, 'attention_fusion_max_rows': 8192},
    # the one-launch rows kernels pulled down to fewer rows, whatever their last round's fill
    'rows>=1024': {'att_rows_min_rows': 1025, 'ffn_rows_min_rows': 1025, 'rows_min_fill': 0},
    'ffn_rows>=1024': {'ffn_rows_min_rows': 1025, 'rows_min_fill': 0, 'att_rows_min_rows': BIG},
    'att_rows>=1024': {'att_rows_min_rows': 1025, 'rows_min_fill': 0, 'ffn_rows_min_rows': BIG},
    'rows_fill>=50': {'rows_min_fill': 50},
    'rows off': {'att_rows_min_rows': BIG, 'ffn_rows_min_rows': BIG},
    # backbone fusions
    'conv23 on': {'conv23_min_pairs': 1},
    'conv23 off': {'conv23_min_pairs': BIG},
    'conv23m on': {'conv23m_min_pairs': 1},
    'conv23m off': {'conv23m_min_pairs': BIG},
    'expand on': {'expand_min_rows': 0},
    'expand off': {'expand_min_rows': BIG},
    'bottleneck on': {'bottleneck_max_pairs': BIG},
    'bottleneck off': {'bottleneck_max_pairs': 0},
    'pos_table on': {'pos_table_min_rows': 0},
}


def flop(b, q):
    return b * 24.641e9 + b * q * 11.273e6


def time_forward(m, img, qs, n):
    for _ in range(3):
        m(img, qs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            m(img, qs)
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', default='1,2,3,4,6,8,12,16,24,32')
    ap.add_argument('--queries', default='1,257,1000')
    ap.add_argument('--sweep', action='store_true')
    ap.add_argument('--only', default='', help='comma-separated candidate names for --sweep (default: all)')
    ap.add_argument('--set', default='', help='knob=value,... applied to every point (instead of the defaults)')
    a = ap.parse_args()
    m = build_model(cotr_amd.default_args()).cuda().eval()
    m.load_state_dict(synth_state_dict(0))
    base = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in a.set.split(',') if kv}
    cands = {k: v for k, v in CANDIDATES.items() if not a.only or k in a.only.split(',')} if a.sweep else {}
    m.reserve(max(int(b) for b in a.pairs.split(',')), max(int(q) for q in a.queries.split(',')))
    print('# forward time and fraction of the fp32-MFMA peak (157.3 TFLOP/s) by batch; knobs: ' + (a.set or 'shipped defaults'))
    for q in [int(x) for x in a.queries.split(',')]:
        for b in [int(x) for x in a.pairs.split(',')]:
            img, qs = synth_inputs(b, q, seed=1)
            img, qs = img.cuda(), qs.cuda()
            n = max(5, min(100, int(60 / (0.8 * b))))
            m.reset_knobs()
            for k, v in base.items():
                m.set_knob(k, v)
            t0 = time_forward(m, img, qs, n)
            line = f'B={b:3d} Q={q:5d}: {t0:8.3f} ms  frac {flop(b, q) / t0 / 1e9 / 157.3:5.3f}  {b * q / t0 * 1e3:10.0f} corr/s'
            res = []
            for name, kn in cands.items():
                m.reset_knobs()
                for k, v in {**base, **kn}.items():
                    m.set_knob(k, v)
                t = time_forward(m, img, qs, n)
                res.append((t, name))
            if res:
                res.sort()
                line += '  |  ' + '  '.join(f'{name} {t / t0 - 1:+.1%}' for t, name in res if abs(t / t0 - 1) >= 0.004)
            print(line, flush=True)
    m.reset_knobs()


if __name__ == '__main__':
    main()
