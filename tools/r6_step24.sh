#!/bin/bash
# Round 6, twenty-fourth GPU session: knob posenc_fold (the query encoding on spare workgroups of the stem launch) - parity, A/B
o=gpurun_out/r6_step24
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -4 > $o/pytest_parity.txt
python tools/ab_inproc.py 1 1000 --rounds 7 --check posenc_fold=0 > $o/ab_posenc_fold_b1_q1000.txt 2>&1
python tools/ab_inproc.py 1 1 --rounds 5 --check posenc_fold=0 > $o/ab_posenc_fold_b1_q1.txt 2>&1
python tools/ab_inproc.py 4 257 --rounds 5 --check posenc_fold=0 > $o/ab_posenc_fold_b4_q257.txt 2>&1
python tools/ab_inproc.py 16 1000 --rounds 5 --check posenc_fold=0 > $o/ab_posenc_fold_b16_q1000.txt 2>&1
python tools/ab_inproc.py 32 1000 --rounds 5 --check posenc_fold=0 > $o/ab_posenc_fold_b32_q1000.txt 2>&1
ls -la $o
