set -u
R=$PWD; OUT=$R/gpurun_out/r14; mkdir -p $OUT
GM=0 DCTTS_TRACE=150 DCTTS_TRACE_FILE=$OUT/decode_trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
cat $OUT/decode_trace.txt | cut -c1-400
