#!/bin/bash
# Everything profiles/r4_final_* (parts a-c) and profiles/r4_split_f16_* (part d: the research path of docs/LABNOTES.md 3e, experimental library) is
# built from, in one go on the GPU box.  usage: tools/final_measure_r4.sh [part a|b|c|d|all]
part=${1:-all}
o=gpurun_out/r4_final
mkdir -p $o
export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = a ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/pytest_gpu.txt
  COTR_HIP_EXPERIMENTAL=1 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/pytest_gpu_experimental_library.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1
  python bench.py > $o/bench.json 2> $o/bench.err
  python bench.py --workload batch256 --steps 3 --warmup 1 --no-cpu-baseline --traffic none > $o/bench_batch256.json 2> $o/bench_batch256.err
  python bench.py --workload dense > $o/bench_dense.json 2> $o/bench_dense.err
  python tools/time_configs.py > $o/time_configs_batched.txt 2>&1
fi
if [ "$part" = all ] || [ "$part" = b ]; then
  for st in 2 1; do
    python bench.py --workload train --stage $st --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage${st}.json
    python bench.py --workload train --stage $st --graphed-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage${st}_graphed.json
  done
  bash tools/prof.sh r4t --workload train --stage 2 --traffic none > /dev/null 2>&1
  cp $(ls gpurun_out/prof_r4t/*kernel_stats.csv gpurun_out/prof_r4t/*/*kernel_stats.csv 2>/dev/null | head -1) $o/train_stage2_rocprofv3_kernel_stats.csv 2>/dev/null
  python tools/profile_train.py 2 200 2 > $o/train_profile_stage2.txt 2>&1
  python tools/bench_engine.py 1000 --config2 > $o/bench_engine.txt 2>&1
fi
if [ "$part" = all ] || [ "$part" = c ]; then
  bash tools/prof.sh r4f > /dev/null 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_r4f/*.db gpurun_out/prof_r4f/*/*.db 2>/dev/null | head -1) --csv $o/kernel_trace_per_launch.csv > $o/kernel_trace_per_launch.txt 2>&1
  cp $(ls gpurun_out/prof_r4f/*kernel_stats.csv gpurun_out/prof_r4f/*/*kernel_stats.csv 2>/dev/null | head -1) $o/rocprofv3_kernel_stats.csv 2>/dev/null
  bash tools/pmc.sh r4f > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    python tools/pmc_summary.py $(ls gpurun_out/pmc_r4f_$c/*counter_collection.csv gpurun_out/pmc_r4f_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  done > $o/pmc_hbm_traffic.txt 2>&1
  python tools/kernel_times.py 1 1000 > $o/kernel_times_hip_events_b1_q1000.txt 2>&1
  python tools/kernel_times.py 32 1000 > $o/kernel_times_hip_events_b32_q1000.txt 2>&1
  bash tools/mfma_util.sh 32 1000 $o/mfma_util_and_traffic_b32_q1000.txt > /dev/null 2>&1
fi
if [ "$part" = all ] || [ "$part" = d ]; then
  export COTR_HIP_EXPERIMENTAL=1
  r=gpurun_out/r4_split_f16
  mkdir -p $r
  python tools/bench_split_f16.py > $r/gemm_configs.txt 2>&1
  python tools/bench_attention_h2.py > $r/attention_kernel.txt 2>&1
  for lv in 1 2 3; do python tools/time_configs.py split_f16=$lv > $r/level${lv}_time_configs_batched.txt 2>&1; done
  python tools/kernel_times.py 32 1000 split_f16=3 > $r/level3_kernel_times_hip_events_b32_q1000.txt 2>&1
  python tools/bench_engine.py 1000 --config2 split_f16=3 > $r/level3_bench_engine.txt 2>&1
  unset COTR_HIP_EXPERIMENTAL
  ls -la $r
fi
ls -la $o
