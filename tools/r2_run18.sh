#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2r; mkdir -p $OUT; export TMPDIR=/tmp
DCTTS_TRACE_FILE=$OUT/trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
tail -3 $OUT/trace.txt
