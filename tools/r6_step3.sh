#!/bin/bash
# Round 6, third GPU session: full GPU test suite (with the end-to-end reference-engine goldens), the batch-axis curve with the
# fill-rule thresholds, the fabric traffic of att_rows with / without the pair-per-XCD placement
o=gpurun_out/r6_step3
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $o/pytest_gpu.txt
python tools/frac_by_batch.py --sweep --only "fused<=4096,fused<=8192,rows_fill>=50,rows off,conv23m on,conv23 off" --set ffn_fusion_max_rows=1024,attention_fusion_max_rows=1024 --pairs 2,3,4,6,8 > $o/frac_by_batch_old_thresholds.txt 2>&1
python tools/frac_by_batch.py > $o/frac_by_batch.txt 2>&1
bash tools/mfma_util.sh 32 1000 $o/mfma_util_b32_q1000_plain_grid.txt > /dev/null 2>&1
bash tools/mfma_util.sh 32 1000 $o/mfma_util_b32_q1000_pair_per_xcd.txt xcd_mapping=33 > /dev/null 2>&1
ls -la $o
