#!/bin/bash
# Round 6, twelfth GPU session: conv23m's uneven-fill gap (17 ... 27 pairs) and bottleneck_max_pairs 5 - forward by batch, parity tests
o=gpurun_out/r6_step12
mkdir -p $o
export TMPDIR=/tmp
python tools/frac_by_batch.py --pairs 4,5,6,16,17,20,24,27,28,32 --queries 1,1000 > $o/frac_by_batch_gap.txt 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -4 > $o/pytest_subset.txt
ls -la $o
