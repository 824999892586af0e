#!/bin/bash
# HBM traffic counters of the benchmark command: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (they do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots), kernel-trace only.
tag=${1:-r1}
export TMPDIR=/tmp
root=$PWD
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$ctr
  mkdir -p $out
  cd /tmp
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o $tag -- python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $PMC_EXTRA > $out/run.log 2>&1
  cd $root
  ls $out | head
done
