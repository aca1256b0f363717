#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2h; mkdir -p $OUT; export TMPDIR=/tmp
DCTTS_TRACE_FILE=$OUT/trace_full.txt timeout 100 python tools/decode_trace.py > $OUT/trace_full.log 2>&1
DCTTS_V3_SKIP=1 DCTTS_TRACE_FILE=$OUT/trace_chainonly.txt timeout 100 python tools/decode_trace.py > $OUT/trace_chainonly.log 2>&1
tail -3 $OUT/trace_full.log
