#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2k; mkdir -p $OUT; export TMPDIR=/tmp
DCTTS_V3_SKIP=1 DCTTS_TRACE_FILE=$OUT/trace_chainonly.txt timeout 100 python tools/decode_trace.py > $OUT/trace_chainonly.log 2>&1
tail -5 $OUT/trace_chainonly.txt
