#!/bin/bash
# rocprofv3 kernel trace of the benchmark command; summaries land in gpurun_out/prof_<tag>/
tag=${1:-r1}
shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out -o $tag -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/run.log 2>&1
cd $OLDPWD
ls -R $out | head -30
