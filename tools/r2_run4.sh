#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2d; mkdir -p $OUT; export TMPDIR=/tmp
for sk in 0 1 2; do for gm in 0 1; do
  DCTTS_V3_SKIP=$sk DCTTS_HOSTTIME=1 DM=3 GM=$gm timeout 120 python tools/decode_time.py > $OUT/time_skip${sk}_gm$gm.log 2>&1
  echo "SKIP=$sk GM=$gm: $(grep text2mel $OUT/time_skip${sk}_gm$gm.log) $(grep -m1 'host enqueue' $OUT/time_skip${sk}_gm$gm.log)"
done; done
timeout 100 python tools/decode_trace.py > $OUT/decode_trace.log 2>&1
cp gpurun_out/decode_trace.txt $OUT/ 2>/dev/null
DCTTS_V3_SKIP=1 timeout 100 python tools/decode_trace.py > $OUT/decode_trace_nobulk.log 2>&1
