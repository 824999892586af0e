#!/bin/bash
# sweeps the kernel-argument placement of the HIP runtime and eager vs graph replay; output -> gpurun_out/exp_launch_overhead.txt
mkdir -p gpurun_out
out=gpurun_out/exp_launch_overhead.txt
: > $out
for k in unset 0 1; do
  for m in eager graph; do
    if [ $k = unset ]; then env -u HIP_FORCE_DEV_KERNARG timeout 300 python tools/exp_launch_overhead.py $m >> $out 2>&1
    else HIP_FORCE_DEV_KERNARG=$k timeout 300 python tools/exp_launch_overhead.py $m >> $out 2>&1; fi
  done
done
cat $out
