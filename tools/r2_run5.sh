#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2e; mkdir -p $OUT; export TMPDIR=/tmp
for c3 in 0 1; do for sk in 0 1; do for gm in 0 1; do
  DCTTS_CHAIN3=${c3} DCTTS_V3_SKIP=${sk} DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_c${c3}_skip${sk}_gm${gm}.log 2>&1
  echo "CHAIN3=${c3} SKIP=${sk} GM=${gm}: $(grep text2mel $OUT/time_c${c3}_skip${sk}_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_c${c3}_skip${sk}_gm${gm}.log)"
done; done; done
