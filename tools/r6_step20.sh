#!/bin/bash
# Round 6, twentieth GPU session: the interpolating pick for row counts between measured entries - off-grid shapes (pair counts and
# query counts without entries) pick against best, and the batch axis at every pair count (knob batch_split off, to see the raw staircase)
o=gpurun_out/r6_step20
mkdir -p $o
export TMPDIR=/tmp
timeout 1500 python tools/mid_batch_cfgs.py 9 11 13 15 18 22 26 30 > $o/off_grid_cfgs_q1000.txt 2>&1
timeout 900 python tools/mid_batch_cfgs.py 5 9 13 18 22 26 30 --q 600 --dec-only > $o/off_grid_cfgs_q600.txt 2>&1
P=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,40,47,48,49,56,63,64
python tools/frac_by_batch.py --pairs $P --queries 1,1000 --set batch_split=0 > $o/frac_every_pair_count_split_off.txt 2>&1
python tools/frac_by_batch.py --pairs 5,9,13,18,22,26,30 --queries 181,600 > $o/frac_off_grid_queries.txt 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -3 > $o/pytest_subset.txt
ls -la $o
