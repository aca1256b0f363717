#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
for e in 0 1 2; do
DCTTS_RC_EXP=$e DCTTS_ROWCHAIN=1 DCTTS_V3_SKIP=1 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/exp$e.log 2>&1
echo "chain only EXP=$e: $(grep text2mel $OUT/exp$e.log)"
done
