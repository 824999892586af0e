#!/bin/bash
# Round 6, first GPU session: (1) parity of the att_rows XCD remap + side stream, (2) their A/Bs, (3) the batch-axis sweep,
# (4) the one-pair tuned table re-checked in situ.   usage: tools/r6_step1.sh
o=gpurun_out/r6_step1
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "not experimental" 2>&1 | tail -5 > $o/pytest_subset.txt
python tools/ab_inproc.py 1 1000 --rounds 7 --check side_stream=1 side_stream=2 side_stream=3 > $o/ab_side_stream_b1_q1000.txt 2>&1
python tools/ab_inproc.py 1 100 --rounds 5 --check side_stream=3 > $o/ab_side_stream_b1_q100.txt 2>&1
python tools/ab_inproc.py 4 257 --rounds 5 --check side_stream=3 > $o/ab_side_stream_b4_q257.txt 2>&1
for sh in "32 1000" "32 1" "16 1000" "64 1000" "4 131072"; do
  set -- $sh
  python tools/ab_inproc.py $1 $2 --rounds 5 --check xcd_mapping=33 > $o/ab_att_rows_xcd_b$1_q$2.txt 2>&1
done
python tools/frac_by_batch.py --sweep > $o/frac_by_batch_sweep.txt 2>&1
python tools/conv_cfgs_in_situ.py 1 > $o/conv_cfgs_in_situ_b1.txt 2>&1
python tools/linear_cfgs_in_situ.py > $o/linear_cfgs_in_situ.txt 2>&1
ls -la $o
