#!/bin/bash
# round 5, second GPU step: att_rows.hip - op tests, A/B, forward A/B, parity
o=gpurun_out/r5c
mkdir -p $o
timeout 300 python -m pytest tests/test_ops_gpu.py -k "att_rows or ffn_rows" -x -q 2>&1 | tail -15 > $o/pytest_rows.txt
timeout 300 python tools/bench_att_rows.py > $o/ab_att_rows.txt 2>&1
timeout 300 python tools/bench_ffn_rows.py 16384 32000 32768 > $o/ab_ffn_rows.txt 2>&1
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -15 > $o/pytest_parity.txt
timeout 600 python tools/time_configs.py > $o/time_on.txt 2>&1
timeout 600 python tools/time_configs.py att_rows_min_rows=1073741824 > $o/time_att_off.txt 2>&1
tail -n 12 $o/*.txt
