#!/bin/bash
# Round 6, seventh GPU session: does hipExtAnyOrderLaunch overlap dependent launches on gfx950 (tools/micro/anyorder_probe.hip), and is
# the measured configuration table still the best pick in the middle of the batch axis (tools/mid_batch_cfgs.py)
o=gpurun_out/r6_step7
mkdir -p $o
export TMPDIR=/tmp
timeout 180 tools/micro/anyorder_probe.exe > $o/anyorder_probe.txt 2>&1
echo "exit $?" >> $o/anyorder_probe.txt
timeout 1500 python tools/mid_batch_cfgs.py 2 3 4 6 8 12 16 24 > $o/mid_batch_cfgs.txt 2>&1
ls -la $o
