#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2t; mkdir -p $OUT; export TMPDIR=/tmp
GM=1 DCTTS_TRACE_FILE=$OUT/trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
tail -50 $OUT/trace.txt
GM=1 DCTTS_V3_SKIP=1 DCTTS_TRACE_FILE=$OUT/trace_nobulk.txt timeout 100 python tools/decode_trace.py > $OUT/trace2.log 2>&1
tail -50 $OUT/trace_nobulk.txt
