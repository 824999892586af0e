#!/bin/bash
# rocprofv3 kernel trace + stats of the benchmark command; outputs land in gpurun_out/prof_<tag>/
tag=${1:-r1}
shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out -o $tag -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/run.log 2>&1
cd $root
ls $out | head -30
