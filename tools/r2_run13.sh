#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2m; mkdir -p $OUT; export TMPDIR=/tmp
for sk in 0 1 2; do
echo "--- SKIP=$sk"
DCTTS_V3_SKIP=$sk DCTTS_PIECETIME=100 DM=3 GM=1 timeout 120 python tools/decode_time.py 2>&1 | grep "frame 10[1-3]\|text2mel" | tail -4
done
cd /tmp
DM=3 GM=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -- python $R/tools/decode_only.py 60 > $OUT/kt3.log 2>&1
cd $R
find $OUT/kt3 -name "*kernel_stats.csv" | head -1 | xargs head -16
