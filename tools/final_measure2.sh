#!/bin/bash
# bench line + the rocprofv3 / PMC / HIP-event profiles of the SAME box in one gpurun call (boxes differ by up to 15 %)
tag=${1:-r2}
o=gpurun_out/${tag}_final
mkdir -p $o
export TMPDIR=/tmp
python bench.py > $o/bench.json 2> $o/bench.err
bash tools/final_measure.sh $tag b
python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --traffic none > $o/bench_after.json 2>/dev/null
