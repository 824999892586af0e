#!/bin/bash
# quick GPU regression + measurement round: op/parity tests, conv phase stamps, in-situ conv times, headline forward time
mkdir -p gpurun_out
tag=${1:-a}
tests=${2:-"tests/test_ops_gpu.py tests/test_parity_gpu.py"}
( timeout 900 python -m pytest $tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r3_${tag}_pytest.txt
timeout 300 python tools/conv_phases.py > gpurun_out/r3_${tag}_conv_phases.txt 2>&1
timeout 300 python tools/conv_cfgs_in_situ.py 1 > gpurun_out/r3_${tag}_conv_in_situ.txt 2>&1
timeout 300 python tools/exp_launch_overhead.py eager > gpurun_out/r3_${tag}_forward.txt 2>&1
tail -8 gpurun_out/r3_${tag}_pytest.txt; cat gpurun_out/r3_${tag}_conv_phases.txt | cut -c1-250; cat gpurun_out/r3_${tag}_conv_in_situ.txt; tail -2 gpurun_out/r3_${tag}_forward.txt
