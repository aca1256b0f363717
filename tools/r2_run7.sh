#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2g; mkdir -p $OUT; export TMPDIR=/tmp
for ev in "" "1"; do for gm in 0 1 2; do
  DCTTS_EV_SYS=$ev DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_ev${ev}_gm${gm}.log 2>&1
  echo "EV_SYS=$ev GM=${gm}: $(grep text2mel $OUT/time_ev${ev}_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_ev${ev}_gm${gm}.log)"
done; done
for sk in 0 1; do for gm in 0 2; do
echo "--- piece times SKIP=$sk GM=$gm"
DCTTS_V3_SKIP=$sk DCTTS_PIECETIME=100 DM=3 GM=$gm timeout 120 python tools/decode_time.py 2>&1 | grep "frame 10[1-6]" | tail -6
done; done
