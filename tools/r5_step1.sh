#!/bin/bash
# round 5, first GPU step: the one-launch FFN block (ffn_rows.hip) - op tests, A/B against the three launches, forward A/B
o=gpurun_out/r5a
mkdir -p $o
python -m pytest tests/test_ops_gpu.py -k "ffn" -x -q 2>&1 | tail -15 > $o/pytest_ffn.txt
python tools/bench_ffn_rows.py > $o/ab_ffn_rows.txt 2>&1
python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -15 > $o/pytest_parity.txt
python tools/time_configs.py > $o/time_on.txt 2>&1
python tools/time_configs.py ffn_rows_min_rows=1073741824 > $o/time_off.txt 2>&1
tail -n 20 $o/*.txt
