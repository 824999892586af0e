#!/bin/bash
# Round 6, ninth GPU session: second pass over the refreshed table (what is still more than 3 % off), the dispatch-knob sweep along the
# batch axis on the new table (do the fill-rule / rows / conv23 thresholds still hold?), training step on the new table
o=gpurun_out/r6_step9
mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python tools/mid_batch_cfgs.py 2 3 4 5 6 7 8 10 12 14 16 20 24 28 > $o/mid_batch_cfgs_q1000_pass2.txt 2>&1
timeout 600 python tools/mid_batch_cfgs.py 2 3 4 5 6 8 10 12 16 20 24 32 --q 257 --dec-only > $o/mid_batch_cfgs_q257_pass2.txt 2>&1
timeout 1500 python tools/frac_by_batch.py --sweep > $o/frac_by_batch_sweep.txt 2>&1
for st in 2 1; do
  python bench.py --workload train --stage $st --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage${st}.json
done
ls -la $o
