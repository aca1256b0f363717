#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for gm in 0 1 2; do
  DCTTS_HOSTTIME=1 DM=3 GM=${gm} timeout 120 python tools/decode_time.py > $OUT/time_gm${gm}.log 2>&1
  echo "GM=${gm}: $(grep text2mel $OUT/time_gm${gm}.log) $(grep -m1 'host enqueue' $OUT/time_gm${gm}.log)"
done
DCTTS_V3_SKIP=1 DM=3 GM=0 timeout 120 python tools/decode_time.py 2>&1 | grep text2mel
DCTTS_PIECETIME=100 DM=3 GM=1 timeout 120 python tools/decode_time.py 2>&1 | grep "frame 10[1-3]" | tail -3
DCTTS_V3_SKIP=1 DCTTS_TRACE_FILE=$OUT/trace_chainonly.txt timeout 100 python tools/decode_trace.py > $OUT/trace_chainonly.log 2>&1
tail -4 $OUT/trace_chainonly.txt
