#!/bin/bash
# Round 6, twenty-seventh GPU session: the implicit data gradient alone (captured step: GPU-bound), interleaved
o=gpurun_out/r6_step27
mkdir -p $o
export TMPDIR=/tmp
for rep in 1 2 3; do
  for d in 0 1; do
    COTR_IMPLICIT_WGRAD=1 COTR_IMPLICIT_DGRAD=$d python bench.py --workload train --stage 2 --graphed-train --steps 40 --warmup 5 2>/dev/null | tail -1 > $o/graphed_wgrad1_dgrad${d}_$rep.json
  done
done
COTR_IMPLICIT_WGRAD=0 COTR_IMPLICIT_DGRAD=0 python bench.py --workload train --stage 2 --graphed-train --steps 40 --warmup 5 2>/dev/null | tail -1 > $o/graphed_wgrad0_dgrad0_1.json
COTR_IMPLICIT_WGRAD=1 COTR_IMPLICIT_DGRAD=1 python bench.py --workload train --stage 1 --graphed-train --steps 40 --warmup 5 2>/dev/null | tail -1 > $o/graphed_stage1.json
ls $o
