#!/bin/bash
# Round 6, twenty-first GPU session: knob batch_split - measure the encode staircase, write csrc/enc_split.inc ON THE BOX, rebuild, and
# take the whole batch axis again (every pair count) with the knob off / on; parity tests of the split passes
o=gpurun_out/r6_step21
mkdir -p $o
export TMPDIR=/tmp
python tools/batch_cost.py --write > $o/batch_cost.txt 2>&1
cp cotr_amd/csrc/enc_split.inc $o/enc_split.inc
python -m cotr_amd.build --experimental > $o/build.txt 2>&1
P=1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,40,47,48,49,56,63,64,65,72,96
python tools/frac_by_batch.py --pairs $P --queries 1,1000 --set batch_split=0 > $o/frac_every_pair_count_split_off.txt 2>&1
python tools/frac_by_batch.py --pairs $P --queries 1,257,1000 > $o/frac_every_pair_count_split_on.txt 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_zoom_engine_gpu.py tests/test_e2e_reference_engines.py -m gpu -q -x 2>&1 | tail -15 > $o/pytest_subset.txt
ls -la $o
