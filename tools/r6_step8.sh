#!/bin/bash
# Round 6, eighth GPU session: the measured configuration table in the MIDDLE of the batch axis - sweep (round-robin timing), fold the
# winners into csrc/gemm_tuned.inc ON THE BOX, rebuild, and measure the forward by batch before / after; parity tests on the new picks
o=gpurun_out/r6_step8
mkdir -p $o
export TMPDIR=/tmp
python tools/frac_by_batch.py > $o/frac_by_batch_before.txt 2>&1
timeout 1500 python tools/mid_batch_cfgs.py 2 3 4 5 6 8 10 12 16 20 24 > $o/mid_batch_cfgs_q1000.txt 2>&1
timeout 600 python tools/mid_batch_cfgs.py 2 3 4 5 6 8 10 12 16 20 24 32 --q 257 --dec-only > $o/mid_batch_cfgs_q257.txt 2>&1
python tools/apply_mid_batch_cfgs.py $o/mid_batch_cfgs_q1000.txt $o/mid_batch_cfgs_q257.txt --min-gain 0.04 > $o/apply.txt 2>&1
cp cotr_amd/csrc/gemm_tuned.inc $o/gemm_tuned.inc
python -m cotr_amd.build --experimental > $o/build.txt 2>&1
python tools/frac_by_batch.py > $o/frac_by_batch_after.txt 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_zoom_engine_gpu.py tests/test_e2e_reference_engines.py -m gpu -q -x 2>&1 | tail -5 > $o/pytest_subset.txt
ls -la $o
