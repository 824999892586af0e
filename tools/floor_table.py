"""Per-launch floor of the ONE-PAIR forward (1 pair x 1000 queries, the metric's configuration): for each of the 94 launches of the
schedule, what the chip could do at best with THIS launch's grid - and how far the measured launch is from it.  CPU-only: reads the
committed measurements.

    python tools/floor_table.py [--names profiles/r5_final_kernel_times_hip_events_b1_q1000.txt]
                                [--trace profiles/r5_final_kernel_trace_per_launch.txt] [--markdown]

Columns (microseconds):
  measured   rocprofv3 kernel duration of the launch in the un-instrumented chain (the chain is back to back: the durations tile the
             forward: 835 of 841 us)
  mfma       FLOP of the launch / 157.3 TFLOP/s (all 256 CUs, every matrix pipe busy all the time)
  mfma@grid  the same at the launch's own grid: a workgroup's FLOP at one CU's share of the peak (0.6145 TFLOP/s) x the rounds its grid
             needs on 256 CUs (a launch with 128 workgroups cannot use more than half the chip; 384 need two rounds)
  ingest     bytes the busiest CU must pull for its workgroups (operand tiles of its tiles over the whole K, weights of a fused block,
             K_h / V_h of an attention tile) / 55 GB/s, the rate one CU sustains when every CU pulls (docs/LABNOTES.md 3c: 51-76 GB/s
             measured, whatever is in flight)
  fixed      4.1 us: what the cheapest launch of this chain costs in situ (head2, 250 workgroups, 0.5 MFLOP: 4.13 us; posenc 4.9;
             ln_reduce 4.8-5.0; an EMPTY dependent chain is 1.7 us per launch, the rest is the previous launch's dirty lines written
             back, first-touch instruction / kernarg fetch, one load -> compute -> store latency chain that nothing overlaps)
  floor      max(mfma@grid, ingest) + fixed  (compute and ingest overlapped perfectly)
"""
import argparse
import collections
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK, CUS, CU_INGEST, FIXED = 157.3e12, 256, 55e9, 4.1
# launch configuration -> (tile rows, tile columns) of csrc/gemm.hip kCfgs (k-split / wave-private tiles contract over the WHOLE K)
TILE = {2: (64, 64), 3: (32, 32), 4: (32, 32), 9: (64, 32), 10: (32, 64), 12: (64, 64), 13: (32, 32), 14: (32, 32), 19: (32, 32),
        22: (32, 16), 23: (32, 16), 24: (32, 16), 25: (32, 16), 26: (128, 128), 27: (128, 64), 30: (32, 16), 31: (32, 16),
        32: (32, 32), 33: (32, 16), 34: (32, 32), 35: (32, 64), 36: (32, 32), 37: (64, 32), 38: (32, 32), 39: (32, 16)}


def gemm(m, n, k, cfg):
    bm, bn = TILE[cfg]
    wgs = -(-m // bm) * (n // bn)
    return 2.0 * m * n * k, wgs, (bm + bn) * k * 4.0


def describe(name):
    """-> (FLOP, workgroups, bytes one workgroup pulls) of a launch from its profiling name."""
    m = re.match(r'(?:conv\dx\d/\d|linear) (\d+)x(\d+)x(\d+) cfg(\d+)', name)
    if m:
        return gemm(*(int(v) for v in m.groups()))
    m = re.match(r'conv1x1/2\+conv1x1/1 (\d+)x(\d+)x(\d+)\+(\d+)x(\d+)x(\d+) cfg(\d+)', name)
    if m:
        v = [int(x) for x in m.groups()]
        f0, w0, b0 = gemm(v[0], v[1], v[2], v[6])
        f1, w1, b1 = gemm(v[3], v[4], v[5], v[6])
        return f0 + f1, w0 + w1, max(b0, b1)
    if name.startswith('stem_pool'):
        return 2.0 * 128 * 256 * 64 * 147, 256, 1.57e6 / 256 * 2.3 + 40e3          # its rows of the image (7x7/2 halo) + the weights
    if name.startswith('bottleneck layer1.0'):
        return 2.0 * 8192 * (64 * 64 + 576 * 64 + 64 * 256 + 64 * 256), 256, 64 * 4 * (64 + 576 + 256 + 256) + 60 * 64 * 4
    if name.startswith('bottleneck'):
        return 2.0 * 8192 * (256 * 64 + 576 * 64 + 64 * 256), 256, 64 * 4 * (256 + 576 + 256) + 60 * 256 * 4
    if name.startswith('attention+oproj enc'):            # 32 queries x 1 head: K_h, V_h, q rows, Wo_h
        return 512 * (2 * 2 * 512 * 256 + 2 * 256 * 256), 128, 2 * 512 * 32 * 4 + 32 * 32 * 4 + 256 * 32 * 4
    if name.startswith('qproj+attention+oproj dec'):      # + the rows to project (tgt, query_pos) and Wq_h
        return 1000 * (2 * 2 * 512 * 256 + 2 * 2 * 256 * 256), 256, 2 * 512 * 32 * 4 + 2 * 32 * 256 * 4 + 2 * 256 * 32 * 4
    m = re.match(r'ffn_fused (\d+) rows x(\d+)', name)
    if m:
        rows, nch = int(m.group(1)), int(m.group(2))
        return rows * 2.0 * 2 * 256 * 1024, -(-rows // 32) * nch, 32 * 256 * 4 + 2 * (1024 // nch) * 256 * 4
    if name.startswith('ln_reduce'):
        return 0.0, 250, 4 * 9 * 1024.0                   # 4 rows x (8 partial slabs + the residual)
    return 0.0, 250, 4096.0                               # posenc, head2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--names', default=os.path.join(ROOT, 'profiles', 'r5_final_kernel_times_hip_events_b1_q1000.txt'))
    ap.add_argument('--trace', default=os.path.join(ROOT, 'profiles', 'r5_final_kernel_trace_per_launch.txt'))
    ap.add_argument('--markdown', action='store_true')
    a = ap.parse_args()
    names = [re.match(r'\s*\d+ (.*?)\s+[\d.]+ us', l).group(1) for l in open(a.names) if re.match(r'\s*\d+ \S.* us$', l)]
    trace = [(re.search(r'grid (\d+)x(\d+)x', l), float(re.search(r'([\d.]+) us$', l).group(1)), l) for l in open(a.trace)
             if re.match(r'\s*\d+ _Z', l)]
    start = next(i for i, t in enumerate(trace) if 'stem_pool' in t[2])       # the rocprofv3 list starts in the middle of a forward
    trace = trace[start:] + trace[:start]
    assert len(names) == len(trace) == 94, (len(names), len(trace))
    rows = collections.OrderedDict()
    tot = collections.Counter()
    for name, (grid, us, _) in zip(names, trace):
        flop, wgs, wg_bytes = describe(name)
        wgs_measured = int(grid.group(1)) * int(grid.group(2))
        rounds = -(-wgs // CUS)
        mfma = flop / PEAK * 1e6
        mfma_grid = (flop / wgs) / (PEAK / CUS) * rounds * 1e6 if flop else 0.0
        ingest = rounds * wg_bytes / CU_INGEST * 1e6
        floor = max(mfma_grid, ingest) + FIXED
        key = re.sub(r'layer1\.[12]', 'layer1.1-2', name)
        r = rows.setdefault(key, dict(n=0, wgs=wgs_measured, flop=0.0, us=0.0, mfma=0.0, mfma_grid=0.0, ingest=0.0, floor=0.0))
        for k, v in (('n', 1), ('flop', flop), ('us', us), ('mfma', mfma), ('mfma_grid', mfma_grid), ('ingest', ingest), ('floor', floor)):
            r[k] += v
            tot[k] += v
    sep = ' | ' if a.markdown else '  '
    head = ['launch (x count)', 'workgroups', 'GFLOP', 'measured', 'mfma', 'mfma@grid', 'ingest', 'fixed', 'floor', 'measured/floor']
    if a.markdown:
        print('| ' + ' | '.join(head) + ' |')
        print('|' + '---|' * len(head))
    else:
        print(f'{head[0]:58s}' + ''.join(f'{h:>11s}' for h in head[1:]))
    for key, r in list(rows.items()) + [('all 94 launches', dict(tot, wgs=0))]:
        n = r['n']
        cells = [f'{key} x{n}' if key != 'all 94 launches' else key, str(r['wgs']) if r['wgs'] else '', f'{r["flop"] / 1e9:.2f}', f'{r["us"]:.1f}', f'{r["mfma"]:.1f}',
                 f'{r["mfma_grid"]:.1f}', f'{r["ingest"]:.1f}', f'{FIXED * n:.1f}', f'{r["floor"]:.1f}', f'{r["us"] / r["floor"]:.2f}']
        if a.markdown:
            print('| ' + ' | '.join(cells) + ' |')
        else:
            print(f'{cells[0]:58s}' + ''.join(f'{c:>11s}' for c in cells[1:]))
    print()
    print(f'sum of the measured launches {tot["us"]:.0f} us; sum of the per-launch floors {tot["floor"]:.0f} us = {tot["flop"] / tot["floor"] / 1e6 / 157.3:.3f} of the fp32-MFMA peak; '
          f'of it fixed {FIXED * 94:.0f} us, matrix work at the launches\' own grids {tot["mfma_grid"]:.0f} us (chip-wide ideal {tot["mfma"]:.0f} us), '
          f'ingest-bound excess {tot["floor"] - FIXED * 94 - tot["mfma_grid"]:.0f} us')


if __name__ == '__main__':
    main()
