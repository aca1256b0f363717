#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
for sk in 1 3; do for gm in 1 2; do
DCTTS_V3_SKIP=$sk DM=3 GM=$gm timeout 120 python tools/decode_time.py > $OUT/sk.log 2>&1
echo "skip $sk graph $gm: $(grep text2mel $OUT/sk.log)"
done; done
DCTTS_HOSTTIME=1 DM=3 GM=1 timeout 120 python tools/decode_time.py 2>&1 | grep "host enqueue" | tail -2
