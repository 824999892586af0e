#!/bin/bash
# Round 6, eleventh GPU session: the refreshed table at the BASELINE shapes, and the dispatch knobs at the batch sizes between the
# measured points (odd pair counts, 40 / 48 / 64 pairs)
o=gpurun_out/r6_step11
mkdir -p $o
export TMPDIR=/tmp
python tools/time_configs.py > $o/time_configs_batched.txt 2>&1
timeout 1800 python tools/frac_by_batch.py --sweep --only "conv23m off,conv23 off,rows off,rows_fill>=50,expand off,bottleneck on,fused<=8192,fused<=2048" --pairs 5,7,10,14,20,24,28,40,48,64 --queries 1,1000 > $o/frac_by_batch_odd_pairs.txt 2>&1
ls -la $o
