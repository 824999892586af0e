#!/bin/bash
# Per kernel of one forward at (B, Q): duration, MFMA pipe utilisation, L2<->fabric bytes and GB/s.  Separate rocprofv3 passes
# (kernel trace + one counter group each).  usage: tools/mfma_util.sh B Q out.txt [KNOB=INT ...]
B=${1:-32}; Q=${2:-1000}; out=${3:-gpurun_out/mfma_util.txt}
export TMPDIR=/tmp
root=$PWD
knobs="${@:4}"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$root/gpurun_out/mfma_util_$i
  rm -rf $d; mkdir -p $d
  cd /tmp
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o u -- python $root/tools/run_forwards.py $B $Q 4 $knobs > $d/run.log 2>&1
  cd $root
  i=$((i+1))
done
python - "$B" "$Q" > $out <<'PY'
import csv, glob, sys, collections
B, Q = sys.argv[1], sys.argv[2]
root = 'gpurun_out'
def load(i):
    f = glob.glob(f'{root}/mfma_util_{i}/**/*counter_collection.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows
def trace(i):
    f = glob.glob(f'{root}/mfma_util_{i}/**/*kernel_trace.csv', recursive=True)[0]
    return {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(f))}
def last_forward(rows):
    # dispatches of the LAST forward: from the last stem_pool kernel on
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    names = {int(r['Dispatch_Id']): r['Kernel_Name'] for r in rows}
    starts = [d for d in ids if 'stem_pool' in names[d]]
    first = starts[-1]
    return [d for d in ids if d >= first], names
r0, r1, r2 = load(0), load(1), load(2)
t0 = trace(0)
ids, names = last_forward(r0)
val = collections.defaultdict(dict)
for r in r0:
    val[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
def per_pos(rows):
    ids_, _ = last_forward(rows)
    v = {int(r['Dispatch_Id']): float(r['Counter_Value']) for r in rows}
    return [v[d] for d in ids_]
fetch, write = per_pos(r1), per_pos(r2)
fam = collections.OrderedDict()
for pos, d in enumerate(ids):
    name = names[d].split('(')[0].replace('void ', '')[:44]
    e = fam.setdefault(name, dict(n=0, ns=0.0, mfma=0.0, gui=0.0, fetch=0.0, write=0.0))
    e['n'] += 1; e['ns'] += t0[d]; e['mfma'] += val[d].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); e['gui'] += val[d].get('GRBM_GUI_ACTIVE', 0.0)
    if pos < len(fetch): e['fetch'] += fetch[pos] * 1024 * 2
    if pos < len(write): e['write'] += write[pos] * 1024
CLOCK_GHZ = 2.38   # shader clock under this load, measured by the un-instrumented probe (tools/clock_settle.py, profiles/r4_shader_clock_probe_and_smi.txt)
print(f'# one forward at B={B}, Q={Q}: per kernel family - launches, time, MFMA pipe utilisation two ways: "util/GUI" = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x')
print('# GRBM_GUI_ACTIVE / 8 XCDs) and "util/time" = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x 2.38 GHz, the probe-measured shader clock) - GRBM_GUI_ACTIVE')
print('# also counts cycles outside the kernel timestamps (it read 4 "GHz" for LayerNorm in round 3: NOT a clock, that column is gone); L2<->fabric bytes')
print('# (FETCH_SIZE x2 + WRITE_SIZE, separate passes) and their rate.  rocprofv3 adds ~2.5 us to short kernels.')
print(f'{"kernel":46s} {"n":>3s} {"us":>9s} {"util/GUI":>9s} {"util/time":>9s} {"MB":>8s} {"GB/s":>7s}')
tot = dict(ns=0.0, mfma=0.0, gui=0.0, b=0.0)
for name, e in sorted(fam.items(), key=lambda kv: -kv[1]['ns']):
    cyc = e['gui'] / 8.0
    util = e['mfma'] / (1024.0 * cyc) if cyc else 0.0
    ghz = e['mfma'] / (1024.0 * e['ns'] * CLOCK_GHZ) if e['ns'] else 0.0
    mb = (e['fetch'] + e['write']) / 1e6
    print(f'{name:46s} {e["n"]:3d} {e["ns"] / 1e3:9.1f} {util:9.3f} {ghz:9.3f} {mb:8.1f} {mb * 1e6 / e["ns"]:7.0f}')
    tot['ns'] += e['ns']; tot['mfma'] += e['mfma']; tot['gui'] += e['gui']; tot['b'] += e['fetch'] + e['write']
cyc = tot['gui'] / 8.0
print(f'{"all kernels":46s} {"":3s} {tot["ns"] / 1e3:9.1f} {tot["mfma"] / (1024.0 * cyc):9.3f} {tot["mfma"] / (1024.0 * tot["ns"] * CLOCK_GHZ):9.3f} {tot["b"] / 1e6:8.1f} {tot["b"] / tot["ns"]:7.0f}')
PY
cat $out
