#!/bin/bash
# SQ counters of one attention variant (separate pass per counter group, kernel-trace only): tools/att_pmc.sh <variant> <tag>
v=${1:-narrow}; tag=${2:-att}
export TMPDIR=/tmp
root=$PWD
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  out=$root/gpurun_out/pmc_${tag}_${v}_$i
  mkdir -p $out
  cd /tmp
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o $tag -- python $root/tools/att_bench.py --variant $v --iters 3 > $out/run.log 2>&1
  cd $root
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$root/gpurun_out/pmc_${tag}_${v}_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'][:40]
            if 'attention' not in kn: continue
            key = (kn, r.get('Grid_Size'))
            acc[key][r['Counter_Name']] += float(r['Counter_Value']); n[(key, r['Counter_Name'])] += 1
        for key, c in acc.items():
            print(key, {k: round(v / n[(key, k)]) for k, v in c.items()})
PY
