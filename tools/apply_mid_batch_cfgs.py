"""Fold the candidates tools/mid_batch_cfgs.py printed ("<== {mode, M, N, K, cfg, cfg_reg},  // t us (pick p)") into csrc/gemm_tuned.inc:
an existing entry of the same shape is replaced, a new shape is appended.  Where two launches share a shape (input_proj / linear2,
q projection / corr_embed) the candidate with the larger saving wins.
    python tools/apply_mid_batch_cfgs.py gpurun_out/r6_step7/mid_batch_cfgs.txt [more outputs ...] [--min-gain 0.04]"""
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inc = os.path.join(root, 'cotr_amd', 'csrc', 'gemm_tuned.inc')
args = sys.argv[1:]
min_gain = 0.04
if '--min-gain' in args:
    i = args.index('--min-gain')
    min_gain = float(args[i + 1])
    del args[i:i + 2]
cand = {}
pat = re.compile(r'<== \{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},\s+// ([\d.]+) us \(pick ([\d.]+)\)')
for path in args:
    for line in open(path):
        m = pat.search(line)
        if not m:
            continue
        mode, M, N, K, cfg, reg = (int(m.group(i)) for i in range(1, 7))
        best, pick = float(m.group(7)), float(m.group(8))
        if best > (1.0 - min_gain) * pick:
            continue
        key = (mode, M, N, K)
        if key not in cand or pick - best > cand[key][3] - cand[key][2]:
            cand[key] = (cfg, reg, best, pick)
lines = open(inc).read().split('\n')
ent = re.compile(r'^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},(.*)$')
seen, out, replaced = set(), [], 0
for l in lines:
    m = ent.match(l)
    if m:
        key = tuple(int(m.group(i)) for i in range(1, 5))
        if key in cand:
            cfg, reg, best, pick = cand[key]
            old_cfg = int(m.group(5))
            l = f'{{{key[0]}, {key[1]}, {key[2]}, {key[3]}, {cfg}, {reg}}},  // {best:.2f} us (round 6, tools/mid_batch_cfgs.py; {pick:.2f} for config {old_cfg})'
            seen.add(key)
            replaced += 1
    out.append(l)
while out and out[-1] == '':
    out.pop()
added = 0
for key in sorted(cand):
    if key in seen:
        continue
    cfg, reg, best, pick = cand[key]
    out.append(f'{{{key[0]}, {key[1]}, {key[2]}, {key[3]}, {cfg}, {reg}}},  // {best:.2f} us (round 6, tools/mid_batch_cfgs.py; {pick:.2f} for the nearest-shape pick)')
    added += 1
open(inc, 'w').write('\n'.join(out) + '\n')
print(f'{replaced} entries replaced, {added} added')
