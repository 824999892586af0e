#!/bin/bash
# Round 6, thirteenth GPU session: the WHOLE batch axis, every pair count from 1 to 40 (and 48, 56, 63, 64, 65): where are the cliffs
# between the measured points?
o=gpurun_out/r6_step13
mkdir -p $o
export TMPDIR=/tmp
python tools/frac_by_batch.py --pairs 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,40,47,48,49,56,63,64,65,72,96,128 --queries 1,1000 > $o/frac_by_batch_every_pair_count.txt 2>&1
ls -la $o
