#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
for rc in 0 1; do for m in 0 4 8 -192 -224; do
  DCTTS_BULK_CUMASK=$m DCTTS_ROWCHAIN=$rc DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/time_cm.log 2>&1
  echo "ROWCHAIN=$rc CUMASK=$m: $(grep text2mel $OUT/time_cm.log)"
done; done
