#!/bin/bash
# Round 6, twenty-eighth GPU session: the trainable bottleneck as one autograd node - tests, goldens, the captured / eager step with and without
o=gpurun_out/r6_step28
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_train_ops_gpu.py tests/test_training_gpu.py -m gpu -q -x 2>&1 | tail -6 > $o/pytest_train.txt
for rep in 1 2 3; do
  for f in 0 1; do
    COTR_BOTTLENECK_FN=$f python bench.py --workload train --stage 2 --graphed-train --steps 40 --warmup 5 2>/dev/null | tail -1 > $o/graphed_bottleneck_fn${f}_$rep.json
  done
done
for f in 0 1; do
  COTR_BOTTLENECK_FN=$f python bench.py --workload train --stage 2 --steps 40 --warmup 5 2>/dev/null | tail -1 > $o/eager_bottleneck_fn${f}.json
done
ls $o
