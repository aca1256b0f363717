#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
DCTTS_PIECETIME=100 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/piece3.log 2>&1
grep "frame 10[2-5]" $OUT/piece3.log | tail -4
DCTTS_V3_SKIP=1 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/skip1_3.log 2>&1
echo "chain only: $(grep text2mel $OUT/skip1_3.log)"
DCTTS_V3_SKIP=2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/skip2_3.log 2>&1
echo "bulk only: $(grep text2mel $OUT/skip2_3.log)"
for cap in 176 192 224; do
DCTTS_BULK_CAP=$cap DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/cap.log 2>&1
echo "cap $cap: $(grep text2mel $OUT/cap.log)"
done
