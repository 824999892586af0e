#!/bin/bash
# Everything profiles/r6_final_* is built from, in one go on the GPU box.   usage: tools/final_measure_r6.sh [a|b|c|all]
part=${1:-all}
o=gpurun_out/r6_final
mkdir -p $o
export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = a ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1
  python bench.py > $o/bench.json 2> $o/bench.err
  python bench.py --workload batch256 --steps 3 --warmup 1 --no-cpu-baseline --traffic none > $o/bench_batch256.json 2> $o/bench_batch256.err
  python bench.py --workload dense > $o/bench_dense.json 2> $o/bench_dense.err
  python tools/time_configs.py > $o/time_configs_batched.txt 2>&1
  python tools/frac_by_batch.py > $o/frac_by_batch.txt 2>&1
fi
if [ "$part" = all ] || [ "$part" = b ]; then
  for st in 2 1; do
    python bench.py --workload train --stage $st --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage${st}.json
    python bench.py --workload train --stage $st --graphed-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $o/bench_train_stage${st}_graphed.json
  done
  python tools/bench_engine.py 1000 --config2 > $o/bench_engine.txt 2>&1
  python tools/profile_flow.py --resample > $o/profile_flow.txt 2>&1
fi
if [ "$part" = all ] || [ "$part" = c ]; then
  bash tools/prof.sh r6f > /dev/null 2>&1
  python tools/prof_summary.py $(ls gpurun_out/prof_r6f/*.db gpurun_out/prof_r6f/*/*.db 2>/dev/null | head -1) --csv $o/kernel_trace_per_launch.csv > $o/kernel_trace_per_launch.txt 2>&1
  cp $(ls gpurun_out/prof_r6f/*kernel_stats.csv gpurun_out/prof_r6f/*/*kernel_stats.csv 2>/dev/null | head -1) $o/rocprofv3_kernel_stats.csv 2>/dev/null
  bash tools/pmc.sh r6f > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    python tools/pmc_summary.py $(ls gpurun_out/pmc_r6f_$c/*counter_collection.csv gpurun_out/pmc_r6f_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  done > $o/pmc_hbm_traffic.txt 2>&1
  python tools/kernel_times.py 1 1000 > $o/kernel_times_hip_events_b1_q1000.txt 2>&1
  python tools/kernel_times.py 32 1000 > $o/kernel_times_hip_events_b32_q1000.txt 2>&1
  bash tools/mfma_util.sh 32 1000 $o/mfma_util_and_traffic_b32_q1000.txt > /dev/null 2>&1
  bash tools/mfma_util.sh 1 1000 $o/mfma_util_and_traffic_b1_q1000.txt > /dev/null 2>&1
  # rocprofv3 kernel stats of the batched forward (the rows kernels are its dominant kernels)
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_r6b -o r6b -- python $OLDPWD/tools/run_forwards.py 32 1000 6 > /dev/null 2>&1; cd $OLDPWD
  cp $(ls gpurun_out/prof_r6b/*kernel_stats.csv gpurun_out/prof_r6b/*/*kernel_stats.csv 2>/dev/null | head -1) $o/rocprofv3_kernel_stats_b32_q1000.csv 2>/dev/null
fi
ls -la $o
