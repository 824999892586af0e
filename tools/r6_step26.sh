#!/bin/bash
# Round 6, twenty-sixth GPU session: the implicit convolution weight gradient - op tests, training goldens, the stage-2 step with / without it
o=gpurun_out/r6_step26
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_train_ops_gpu.py tests/test_training_gpu.py -m gpu -q -x 2>&1 | tail -6 > $o/pytest_train.txt
for rep in 1 2; do
  COTR_IMPLICIT_WGRAD=0 COTR_IMPLICIT_DGRAD=0 python bench.py --workload train --stage 2 --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage2_im2col_$rep.json
  COTR_IMPLICIT_WGRAD=1 python bench.py --workload train --stage 2 --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage2_implicit_$rep.json
done
COTR_IMPLICIT_WGRAD=0 COTR_IMPLICIT_DGRAD=0 python bench.py --workload train --stage 2 --graphed-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage2_graphed_im2col.json
COTR_IMPLICIT_WGRAD=1 python bench.py --workload train --stage 2 --graphed-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage2_graphed_implicit.json
ls -la $o
