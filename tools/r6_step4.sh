#!/bin/bash
# Round 6, fourth GPU session: the research library's suite (failed in step 3: output captured this time), the full GPU suite,
# the dense pass phase by phase, the engine bench
o=gpurun_out/r6_step4
mkdir -p $o
export TMPDIR=/tmp
COTR_HIP_EXPERIMENTAL=1 python -m pytest tests/test_experimental_gpu.py -m gpu -q -x 2>&1 | tail -80 > $o/pytest_experimental.txt
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $o/pytest_gpu.txt
python tools/profile_flow.py > $o/profile_flow.txt 2>&1
python tools/profile_flow.py --resample > $o/profile_flow_resample.txt 2>&1
python tools/bench_engine.py 1000 > $o/bench_engine.txt 2>&1
python tools/ab_inproc.py 32 1000 --rounds 5 --check xcd_mapping=33 > $o/ab_att_rows_xcd_b32_q1000.txt 2>&1
python tools/ab_inproc.py 4 131072 --rounds 3 --iters 5 --check xcd_mapping=33 > $o/ab_att_rows_xcd_b4_q131072.txt 2>&1
ls -la $o
