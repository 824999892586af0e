#!/bin/bash
# round 5, GPU step 3: forward A/B with the two "rows" kernels (ffn_rows, att_rows), parity, engine
o=gpurun_out/r5d
mkdir -p $o
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -5 > $o/pytest_parity_ops.txt
timeout 300 python tools/bench_att_rows.py > $o/ab_att_rows.txt 2>&1
timeout 600 python tools/time_configs.py > $o/time_on.txt 2>&1
timeout 600 python tools/time_configs.py att_rows_min_rows=1073741824 > $o/time_att_off.txt 2>&1
timeout 600 python tools/time_configs.py att_rows_min_rows=1073741824 ffn_rows_min_rows=1073741824 > $o/time_both_off.txt 2>&1
tail -n 9 $o/*.txt | cut -c1-200
