#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_full.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_full.log
tail -5 $OUT/pytest_full.log
DCTTS_PIECETIME=100 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/piece3.log 2>&1
grep "frame 10[2-5]" $OUT/piece3.log | tail -4
DCTTS_V3_SKIP=2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/sv.log 2>&1
echo "bulk only mode 3: $(grep text2mel $OUT/sv.log)"
