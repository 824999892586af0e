#!/bin/bash
# Round 6, tenth GPU session: the same table check at the TOP of the batch axis (32 / 48 / 64 pairs per encode pass; 256-pair batches run as
# four 64-pair passes) and at the dense pass's decode chunk
o=gpurun_out/r6_step10
mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python tools/mid_batch_cfgs.py 32 48 64 > $o/mid_batch_cfgs_b32_48_64.txt 2>&1
ls -la $o
