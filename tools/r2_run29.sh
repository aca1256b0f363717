#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
DCTTS_HOSTTIME=1 DM=3 GM=1 timeout 120 python tools/decode_time.py 2>&1 | grep "host enqueue\|text2mel"
DCTTS_HOSTTIME=1 DCTTS_V3_SKIP=3 DM=3 GM=1 timeout 120 python tools/decode_time.py 2>&1 | grep "host enqueue\|text2mel"
nproc; cat /proc/cpuinfo | grep "model name" | head -1
