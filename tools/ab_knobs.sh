#!/bin/bash
# A/B of tuning knobs on the headline workload: ms per forward with each setting, interleaved, 3 rounds
# usage: tools/ab_knobs.sh <out file> [knob=val ...]      (default set: the small-row fusions)
out=${1:-gpurun_out/ab_knobs.txt}
shift
variants=("$@")
if [ ${#variants[@]} -eq 0 ]; then variants=("attention_fusion_max_rows=0" "head_fusion_max_rows=2048" "dual_conv=0" "ks3=0"); fi
: > $out
for round in 1 2 3; do
  for v in default "${variants[@]}"; do
    if [ "$v" = default ]; then a=""; else a="--set $v"; fi
    ms=$(python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --traffic none $a 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round  $v  $ms ms" >> $out
  done
done
cat $out
