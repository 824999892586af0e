#!/bin/bash
# Round 6, seventeenth GPU session: decode passes of ONE round of the chip (16384 query rows = 256 tiles of 64) against the shipped 32768
o=gpurun_out/r6_step17
mkdir -p $o
export TMPDIR=/tmp
python tools/time_configs.py > $o/time_configs_dec_rows_32768.txt 2>&1
sed -i 's/constexpr int DEC_ROWS = 32768;/constexpr int DEC_ROWS = 16384;/' cotr_amd/csrc/api.hip
python -m cotr_amd.build > $o/build.txt 2>&1
python tools/time_configs.py > $o/time_configs_dec_rows_16384.txt 2>&1
python tools/frac_by_batch.py --pairs 16,24,32,48,64 --queries 257,1000,2048 > $o/frac_dec_rows_16384.txt 2>&1
sed -i 's/constexpr int DEC_ROWS = 16384;/constexpr int DEC_ROWS = 32768;/' cotr_amd/csrc/api.hip
python -m cotr_amd.build >> $o/build.txt 2>&1
python tools/frac_by_batch.py --pairs 16,24,32,48,64 --queries 257,1000,2048 > $o/frac_dec_rows_32768.txt 2>&1
ls -la $o
