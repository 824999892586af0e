#!/bin/bash
# Round 6, fifth GPU session: full GPU suite after the dispatch / research-predicate fixes, the producer-side slab count A/B (FFN),
# a default bench.py line (by_batch)
o=gpurun_out/r6_step5
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $o/pytest_gpu.txt
python tools/ab_inproc.py 1 1000 --rounds 7 --check ffn_fused_max_chunks=8 ffn_fused_max_chunks=4 attention_fused_splits=8 > $o/ab_slab_count_b1_q1000.txt 2>&1
python tools/ab_inproc.py 1 1 --rounds 5 ffn_fused_max_chunks=8 > $o/ab_slab_count_b1_q1.txt 2>&1
python bench.py > $o/bench.json 2> $o/bench.err
ls -la $o
