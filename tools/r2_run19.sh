#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r2s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_vs_oracle_loop or end_of_text or golden_config1 or ragged" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for cfg in "0 2" "1 2" "1 4"; do set -- $cfg
  DCTTS_MLP=$1 DCTTS_MLP_ROWS=$2 DM=3 GM=1 timeout 120 python tools/decode_time.py > $OUT/time_m$1_$2.log 2>&1
  echo "MLP=$1 ROWS=$2: $(grep text2mel $OUT/time_m$1_$2.log)"
done
DCTTS_TRACE_FILE=$OUT/trace.txt timeout 100 python tools/decode_trace.py > $OUT/trace.log 2>&1
tail -2 $OUT/trace.txt
