#!/bin/bash
# Round 6, sixteenth GPU session: the training step's GEMM shapes against the table, round-robin
o=gpurun_out/r6_step16
mkdir -p $o
export TMPDIR=/tmp
timeout 1200 python tools/train_cfgs.py 2 > $o/train_cfgs_stage2.txt 2>&1
ls -la $o
