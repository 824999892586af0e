#!/bin/bash
# Round 6, second GPU session: the A/Bs of step 1 (argument parsing fixed) + per-launch times along the batch axis
o=gpurun_out/r6_step2
mkdir -p $o
export TMPDIR=/tmp
python tools/ab_inproc.py 1 1000 --rounds 7 --check side_stream=1 side_stream=2 side_stream=3 > $o/ab_side_stream_b1_q1000.txt 2>&1
python tools/ab_inproc.py 1 100 --rounds 5 --check side_stream=3 > $o/ab_side_stream_b1_q100.txt 2>&1
python tools/ab_inproc.py 4 257 --rounds 5 --check side_stream=3 > $o/ab_side_stream_b4_q257.txt 2>&1
for sh in "32 1000" "32 1" "16 1000" "64 1000" "4 131072"; do
  set -- $sh
  python tools/ab_inproc.py $1 $2 --rounds 5 --check xcd_mapping=33 > $o/ab_att_rows_xcd_b$1_q$2.txt 2>&1
done
for b in 2 4 8 12 16 24; do
  python tools/kernel_times.py $b 1000 > $o/kernel_times_b${b}_q1000.txt 2>&1
done
ls -la $o
