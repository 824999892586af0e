#!/bin/bash
# Everything the round's profiles/ directory is built from, in one go on the GPU box.  usage: tools/final_measure.sh <tag> [part]
tag=${1:-r2}; part=${2:-all}
o=gpurun_out/${tag}_final
mkdir -p $o
export TMPDIR=/tmp
if [ "$part" = all ] || [ "$part" = a ]; then
  python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1
  python bench.py > $o/bench.json 2> $o/bench.err
  python bench.py --workload batch256 --steps 3 --warmup 1 --no-cpu-baseline --traffic none > $o/bench_batch256.json 2> $o/bench_batch256.err
  python bench.py --workload train --no-cpu-baseline --traffic none > $o/bench_train.json 2> $o/bench_train.err
  python tools/time_configs.py > $o/time_configs_batched.txt 2>&1
fi
if [ "$part" = all ] || [ "$part" = b ]; then
  bash tools/prof.sh ${tag}f > /dev/null 2>&1
  python tools/prof_summary.py gpurun_out/prof_${tag}f/*/${tag}f_results.db --csv $o/kernel_trace_per_launch.csv > $o/kernel_trace_per_launch.txt 2>&1 || \
    python tools/prof_summary.py $(ls gpurun_out/prof_${tag}f/*.db gpurun_out/prof_${tag}f/*/*.db 2>/dev/null | head -1) --csv $o/kernel_trace_per_launch.csv > $o/kernel_trace_per_launch.txt 2>&1
  cp $(ls gpurun_out/prof_${tag}f/*kernel_stats.csv gpurun_out/prof_${tag}f/*/*kernel_stats.csv 2>/dev/null | head -1) $o/rocprofv3_kernel_stats.csv 2>/dev/null
  bash tools/pmc.sh ${tag}f > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    python tools/pmc_summary.py $(ls gpurun_out/pmc_${tag}f_$c/*counter_collection.csv gpurun_out/pmc_${tag}f_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  done > $o/pmc_hbm_traffic.txt 2>&1
  python tools/kernel_times.py 1 1000 > $o/kernel_times_hip_events_b1_q1000.txt 2>&1
  python tools/kernel_times.py 32 1 > $o/kernel_times_hip_events_b32_q1.txt 2>&1
  python tools/kernel_times.py 32 1000 > $o/kernel_times_hip_events_b32_q1000.txt 2>&1
  python tools/kernel_times.py 1 32768 > $o/kernel_times_hip_events_b1_q32768.txt 2>&1
  python tools/att_bench.py > $o/attention_variants.txt 2>&1
  bash tools/ab_knobs.sh $o/ab_knobs.txt attention_fusion_max_rows=0 dual_conv=0 ks3=0 conv_patch=0 head_fusion_max_rows=2048 attention_fused_splits=8 > /dev/null 2>&1
  python tools/profile_train.py > $o/train_profile.txt 2>&1
fi
ls -la $o
