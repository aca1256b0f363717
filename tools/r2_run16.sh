#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2p; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "176 16" "160 16" "144 16" "128 16" "112 16" "192 16" "176 6" "144 6" "128 6"; do set -- $cfg
  DCTTS_BULK_CAP=$1 DCTTS_BULK3_SMALL=$2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/t_$1_$2.log 2>&1
  echo "CAP=$1 SMALL=$2: $(grep text2mel $OUT/t_$1_$2.log)"
done
