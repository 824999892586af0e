#!/bin/bash
# Round 6, thirtieth GPU session: rocprofv3 kernel stats of the forward in the middle of the batch axis (4 / 8 / 16 pairs x 1000 queries)
o=gpurun_out/r6_step30
mkdir -p $o
export TMPDIR=/tmp
for b in 4 8 16; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_r6_b$b -o r6b$b -- python $OLDPWD/tools/run_forwards.py $b 1000 8 > /dev/null 2>&1)
  cp $(ls gpurun_out/prof_r6_b$b/*kernel_stats.csv gpurun_out/prof_r6_b$b/*/*kernel_stats.csv 2>/dev/null | head -1) $o/rocprofv3_kernel_stats_b${b}_q1000.csv 2>/dev/null
  python tools/kernel_times.py $b 1000 > $o/kernel_times_hip_events_b${b}_q1000.txt 2>&1
done
ls -la $o
