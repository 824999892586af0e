#!/bin/bash
# Round-3 training measurements (GPU box): bench lines (stage 1 / 2, eager / captured graph), A/B of GradSink / FusedAdam,
# attention kernels per shape, torch-profiler tables.  Output: gpurun_out/r3_train/
out=gpurun_out/r3_train; mkdir -p $out
for st in 1 2; do
  python bench.py --workload train --stage $st --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_train_stage${st}.json
  python bench.py --workload train --stage $st --graphed-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_train_stage${st}_graphed.json
  python bench.py --workload train --stage $st --no-grad-sink --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_train_stage${st}_no_sink.json
done
python tools/ab_grad_sink.py 2>&1 | grep -v amdgpu > $out/ab_grad_sink.txt
python tools/att_train_bench.py 2>&1 | grep -v "amdgpu\|Warn\|warn" > $out/att_train_bench.txt
python tools/profile_train.py 1 200 2 > $out/train_profile_stage1.txt 2>&1
python tools/profile_train.py 2 200 2 > $out/train_profile_stage2.txt 2>&1
for f in $out/bench_train_*.json; do python -c "import sys,json; d=json.loads(open('$f').read()); print('$f', d['ms_per_step'], d['value'])"; done
