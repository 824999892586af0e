"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per (kernel, grid) launch count per step and
average duration, in launch order of the last traced step.  Writes a CSV next to stdout output.

    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db [--steps 20] [--csv profiles/x.csv]
"""
import argparse
import csv
import re
import sqlite3
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '')
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--csv', default=None)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = c.execute(f'select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, '
                     f'd.workgroup_size_x, s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size '
                     f'from {kd} d join {ks} s on d.kernel_id = s.id order by d.start').fetchall()
    # keep only our kernels (drop torch fill/copy kernels of the setup)
    ours = [r for r in rows if any(k in r[0] for k in ('gemm_kernel', 'gemm_ks_kernel', 'attention', 'attention_kernel', 'layernorm_kernel',
                                                       'posenc_kernel', 'maxpool_kernel', 'head2_kernel', 'fused', 'ln_reduce', 'stem_pool', 'gemm_big', 'gemm_wp', 'gemm_ws', 'bottleneck', 'dual_kernel', 'dec_head'))]
    # one stem convolution (gemm_kernel<2,2,2,1,GEMM_STEM>) per forward: robust against extra forwards in the command
    n_fwd = sum(1 for r in ours if 'stem_pool' in r[0] or
                ('gemm_kernel' in r[0] and ('Li2ELi2ELi2ELi1ELi2E' in r[0] or '<2, 2, 2, 1, 2>' in r[0])))
    per_step = len(ours) // n_fwd if n_fwd else len(ours)
    ours = ours[: per_step * (a.steps + a.warmup)]
    print(f'{len(rows)} dispatches, {len(ours)} ours, {per_step} per step')
    steady = ours[a.warmup * per_step:]
    agg = OrderedDict()
    for i, r in enumerate(steady):
        key = (i % per_step,)
        e = agg.setdefault(key, {'name': short(r[0]), 'grid': (r[3] // r[6], r[4], r[5]), 'vgpr': r[7], 'agpr': r[8],
                                 'lds': r[9], 'n': 0, 'ns': 0})
        e['n'] += 1
        e['ns'] += r[2] - r[1]
    # gaps between consecutive kernels of the last step
    last = steady[-per_step:]
    busy = sum(r[2] - r[1] for r in last)
    span = last[-1][2] - last[0][1]
    out = []
    fam = defaultdict(float)
    for key, e in agg.items():
        us = e['ns'] / e['n'] / 1e3
        out.append([key[0], e['name'], 'x'.join(map(str, e['grid'])), e['vgpr'], e['agpr'], e['lds'], f'{us:.2f}'])
        fam[e['name'].split('<')[0] + ('<' + e['name'].split('<')[1][:12] if '<' in e['name'] else '')] += us
    for o in out:
        print('%3d %-60s grid %-12s vgpr %3s agpr %3s lds %6s  %8s us' % tuple(o))
    print(f'last step: kernel busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us')
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f'  {k:50s} {v:9.1f} us/step')
    if a.csv:
        with open(a.csv, 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['launch_index', 'kernel', 'grid_workgroups', 'arch_vgpr', 'accum_vgpr', 'lds_bytes', 'avg_us'])
            w.writerows(out)
            w.writerow(['#', f'last step busy_us={busy / 1e3:.1f} span_us={span / 1e3:.1f}', '', '', '', '', ''])


if __name__ == '__main__':
    main()
